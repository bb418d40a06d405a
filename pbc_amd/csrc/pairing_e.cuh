// pairing_e.cuh -- Type E (ordinary curve y^2 = x^3 + ax + b over a 1020-bit F_q, embedding degree
// k = 1: G1 = G2 = E(F_q)[r], GT = F_q) Tate pairing, one pairing per lane.
//
// Computes the same GT value as the reference's e_pairing (ecc/e_param.c:472-483):
//     e(P, Q) = ( f_{r,P}(Q + R) / f_{r,P}(R) )^((q-1)/r)
// with an auxiliary point R.  The reference draws R at random when the pairing is initialised
// (curve_set_gen_no_cofac, e_param.c:869-870); the value of the Tate pairing does not depend on
// it, so this engine derives its own R once per parameter set (the curve point with the smallest
// x >= 1, found by a single-lane kernel: Euler criterion + Tonelli-Shanks).  Re-derived for a GPU:
//   * plain double-and-add over the bits of r (any r; the reference's e_miller_proj :64-300 is
//     specialised to r = 2^exp2 +- 2^exp1 +- 1 and keeps a saved copy of the state at 2^exp1);
//   * Jacobian V, line and vertical values scaled by factors common to both evaluation points, so
//     they cancel in numerator/denominator; with k = 1 the verticals do not vanish under the final
//     power and are kept (do_vertical, e_param.c:141-148);
//   * numerator and denominator are accumulated separately, also across the terms of a product
//     (generic_prod_pairings, ecc/pairing.c:35-46: the product of the pairings): one inversion and
//     one (q-1)/r power per lane;
//   * round 5: the steps run on fused products (fp.cuh: no sum or difference through memory on its own), R's
//     x-coordinate is a SMALL INTEGER s by construction, so "ZZ x_R" is s ZZ by additions inside the product that uses
//     it (23 instead of 25 products per doubling step, and the two line values as two-product sums with one reduction each), and the (q-1)/r power runs on a sliding window over the
//     860-bit exponent, which is the same for every lane (180 instead of 430 products besides the squarings).
#pragma once
#include "fp.cuh"

namespace pbc {

constexpr int NE_MAX = 34;
struct EConst {
  uint32_t A[NE_MAX], B[NE_MAX];       // curve coefficients (Montgomery form)
  uint32_t Rx[NE_MAX], Ry[NE_MAX];     // auxiliary point R (Montgomery form)
  uint32_t r[8];                       // group order
  uint32_t phik[NE_MAX];               // (q - 1)/r (e_param.c:857-860)
  int rbits, phikbits;
  int rxs;                             // x_R as an integer (e_init_lane: the smallest x >= 1 on the curve), 1..255
};
static_assert(sizeof(EConst) <= KOFF_XS - KOFF_TYPE, "constant block layout");
#define c_e (pbc::kconst<pbc::EConst, pbc::KOFF_TYPE>())
// host-supplied integers for the one-time search of R
struct ERaw {
  uint32_t a[NE_MAX], b[NE_MAX];       // canonical curve coefficients
  uint32_t half[NE_MAX];               // (q - 1)/2                (Euler criterion)
  uint32_t t[NE_MAX];                  // odd t with q - 1 = 2^s t (Tonelli-Shanks)
  uint32_t t1h[NE_MAX];                // (t + 1)/2
  int halfbits, tbits, t1hbits, s;
};

template <int N>
PBC_DEV fp<N> ek(const uint32_t *w) { fp<N> r; fp_set<N>(r, w); return r; }

template <int N>
PBC_DEV void e_pow(fp<N> &r, const fp<N> &a, const uint32_t *e, int bits) {
  fp<N> acc;
  fp_set<N>(acc, fpk<N>().one);
  for (int i = bits - 1; i >= 0; i--) {
    fp_sqr<N>(acc, acc);
    if ((e[i >> 5] >> (i & 31)) & 1) fp_mul<N>(acc, acc, a);
  }
  r = acc;
}
template <int N>
PBC_DEV bool e_on_curve(const fp<N> &x, const fp<N> &y) {
  fp<N> t0, t1;
  fp_sqr<N>(t0, x);
  fp_add<N>(t0, t0, ek<N>(c_e.A));
  fp_mul<N>(t0, t0, x);
  fp_add<N>(t0, t0, ek<N>(c_e.B));
  fp_sqr<N>(t1, y);
  return fp_eq<N>(t0, t1);
}

// The running point.  ZZ = Z^2 and the steps' scratch element t0 carry a quarter of a step's element moves (20 of 76):
// they live where the caller puts them -- two LDS slots per lane on the device for the 33-word fields (the products
// take generic pointers), the lane's private memory otherwise.
template <int N>
struct ejac {
  fp<N> X, Y, Z;
  fp<N> *zz, *t0;
};
// a lane's two hot elements in LDS: 72 words per lane (16-byte aligned elements at words 0 and 36)
constexpr int kWideHotWords = 72;
template <int N>
struct ekon { fp<N> A, x2, y2; };      // the uniform operands of a step, copied from the constant block once per lane

// x^e for an exponent all lanes share: sliding window over odd powers x, x^3, ..., x^15 (control flow is uniform)
template <int N>
PBC_DEV void e_pow_win(fp<N> &r, const fp<N> &a, const uint32_t *e, int bits, fp<N> *slot) {
  fp<N> tab[8], acc_private;
  fp<N> &acc = slot ? *slot : acc_private;   // read and written by every product: an LDS slot on the device
  {
    fp<N> a2;
    tab[0] = a;
    fp_sqrx<N>(a2, a);
    for (int i = 1; i < 8; i++) fp_mulx<N>(tab[i], tab[i - 1], a2);
  }
  fp_set<N>(acc, fpk<N>().one);
  bool started = false;
  int i = bits - 1;
  while (i >= 0) {
    if (!((e[i >> 5] >> (i & 31)) & 1)) {
      if (started) fp_sqrx<N>(acc, acc);
      i--;
      continue;
    }
    int lo = i >= 3 ? i - 3 : 0;       // bits i .. lo with bit lo set: an odd window of at most four bits
    while (!((e[lo >> 5] >> (lo & 31)) & 1)) lo++;
    int w = 0;
    for (int j = i; j >= lo; j--) w = 2 * w + (int) ((e[j >> 5] >> (j & 31)) & 1);
    if (started) {
      for (int j = i; j >= lo; j--) fp_sqrx<N>(acc, acc);
      fp_mulx<N>(acc, acc, tab[w >> 1]);
    } else {
      acc = tab[w >> 1];
      started = true;
    }
    i = lo - 1;
  }
  r = acc;
}

// n <- n^2 l(S1) v(S2),  d <- d^2 l(S2) v(S1)   for S1 = Q + R (per lane), S2 = R = (s, y2) (uniform)
//   tangent at V scaled by 2 Y Z^3:  l(S) = (Z3 ZZ) ys - 2 Y^2 - M (ZZ xs - X),  M = 3X^2 + a Z^4, Z3 = 2YZ
//   vertical at 2V scaled by Z3^2:   v(S) = Z3^2 xs - X3
// 10 S + 9 M + 2 two-product sums (one reduction each): 21 fused calls.
template <int N>
PBC_DEV void e_double_step(fp<N> &n, fp<N> &d, ejac<N> &V, const fp<N> &x1, const fp<N> &y1, const ekon<N> &K) {
  using namespace fx;
  const int ks = b_times(c_e.rxs);
  fp<N> XX, YY, M, W, l1, l2, S;
  fp<N> &ZZ = *V.zz, &t0 = *V.t0;
  fp_sqrx<N>(n, n);
  fp_sqrx<N>(d, d);
  fp_sqrx<N>(XX, V.X);
  fp_sqrx<N>(YY, V.Y);
  fp_sqrx<N>(t0, ZZ);
  fp_mulx<N>(M, C1_ADD | c1_sh(1) | C2_ADD, t0, t0, K.A, K.A, XX, XX);                   // M = a Z^4 + 2X^2 + X^2
  fp_sqrx<N>(V.Z, A_ADD | C1_SUB | C2_SUB, V.Y, V.Z, YY, ZZ);                          // Z3 = 2YZ, in place
  fp_mulx<N>(W, V.Z, ZZ);
  fp_mulx<N>(t0, C1_SUB, ZZ, ZZ, x1, x1, V.X, V.X);
#if PBC_WIDE_NO_SOP                    // A/B: every product with its own reduction (profiles/r05_ab_wide.txt block 13)
  fp_mulx<N>(t0, M, t0);
  fp_mulx<N>(l1, C1_SUB | c1_sh(1) | C2_SUB, W, W, y1, y1, YY, t0);
  fp_mulx<N>(t0, B_SUB | ks, M, M, ZZ, V.X, M, M);
  fp_mulx<N>(l2, C1_SUB | c1_sh(1) | C2_SUB, W, W, K.y2, K.y2, YY, t0);
#else
  fp_sopx<N>(l1, NEG2 | C1_SUB | c1_sh(1), W, y1, M, t0, t0, YY, YY);                    // l(S1) = W y1 - M (ZZ x1 - X) - 2Y^2: two products, one reduction
  fp_sopx<N>(l2, NEG2 | B_SUB | ks | C1_SUB | c1_sh(1), W, K.y2, M, ZZ, V.X, YY, YY);    // l(S2) = W y2 - M (s ZZ - X) - 2Y^2
#endif
  fp_mulx<N>(n, n, l1);
  fp_mulx<N>(d, d, l2);
  fp_mulx<N>(S, dbl(2), V.X, V.X, YY, YY, YY, YY);                                       // 4XY^2
  fp_sqrx<N>(t0, YY);                                                                    // Y^4
  fp_sqrx<N>(V.X, C1_SUB | c1_sh(1), M, M, S, S);                                        // X3 = M^2 - 2S
  fp_mulx<N>(V.Y, B_SUB | C1_SUB | c1_sh(3), M, M, S, V.X, t0, t0);                      // Y3 = M (S - X3) - 8Y^4
  fp_sqrx<N>(ZZ, V.Z);
  fp_mulx<N>(n, B_SUB | ks, n, n, ZZ, V.X, n, n);                                      // v(S2) = s ZZ - X3
  fp_mulx<N>(t0, C1_SUB, ZZ, ZZ, x1, x1, V.X, V.X);
  fp_mulx<N>(d, d, t0);
}
// chord through V and the affine P scaled by Z3 = Z H:  l(S) = (ys - yP) Z3 - R' (xs - xP),
// H = xP Z^2 - X, R' = yP Z^3 - Y;  then V <- V + P and the verticals at the new V
template <int N>
PBC_DEV void e_add_step(fp<N> &n, fp<N> &d, ejac<N> &V, const fp<N> &xP, const fp<N> &yP, const fp<N> &x1,
                        const fp<N> &y1, const ekon<N> &K) {
  using namespace fx;
  const int ks = b_times(c_e.rxs);
  fp<N> H, Rr, HH, HHH;
  fp<N> &ZZ = *V.zz, &t0 = *V.t0;
  fp_mulx<N>(H, C1_SUB, xP, xP, ZZ, ZZ, V.X, V.X);
  fp_mulx<N>(t0, V.Z, ZZ);
  fp_mulx<N>(Rr, C1_SUB, yP, yP, t0, t0, V.Y, V.Y);
  fp_mulx<N>(V.Z, V.Z, H);                                                               // Z3, in place
  fp_mulx<N>(t0, A_SUB, y1, yP, V.Z, V.Z, V.Z, V.Z);
  fp_mulx<N>(t0, A_SUB | C1_ADD, xP, x1, Rr, Rr, t0, t0);                                // (y1 - yP) Z3 - (x1 - xP) R'
  fp_mulx<N>(n, n, t0);
  fp_mulx<N>(t0, A_SUB, K.y2, yP, V.Z, V.Z, V.Z, V.Z);
  fp_mulx<N>(t0, A_SUB | C1_ADD, xP, K.x2, Rr, Rr, t0, t0);
  fp_mulx<N>(d, d, t0);
  fp_sqrx<N>(HH, H);
  fp_mulx<N>(HHH, HH, H);
  fp_mulx<N>(t0, V.X, HH);
  fp_sqrx<N>(V.X, C1_SUB | C2_SUB | c2_sh(1), Rr, Rr, HHH, t0);                          // X3 = R'^2 - H^3 - 2 X H^2
  fp_mulx<N>(HHH, V.Y, HHH);
  fp_mulx<N>(V.Y, B_SUB | C1_SUB, Rr, Rr, t0, V.X, HHH, HHH);                            // Y3 = R' (X H^2 - X3) - Y H^3
  fp_sqrx<N>(ZZ, V.Z);
  fp_mulx<N>(n, B_SUB | ks, n, n, ZZ, V.X, n, n);
  fp_mulx<N>(t0, C1_SUB, ZZ, ZZ, x1, x1, V.X, V.X);
  fp_mulx<N>(d, d, t0);
}

// numerator and denominator of f_{r,P}(Q+R) / f_{r,P}(R) for one lane (n = d = 1 on entry).
// G1, G2 bytes: x||y.  Returns false when an input deserialises to O (curve_from_bytes,
// ecc/curve.c:609-623).
template <int N>
PBC_DEV bool e_miller_lane(fp<N> &n, fp<N> &d, const uint8_t *g1, const uint8_t *g2, uint32_t *lds_q,
                           int lds_stride) {
  using namespace fx;
  const int NB = fq_bytes<N>();
  fp<N> one, xP, yP, x1, y1;
  ekon<N> K;
  K.A = ek<N>(c_e.A);
  K.x2 = ek<N>(c_e.Rx);
  K.y2 = ek<N>(c_e.Ry);
  fp_set<N>(one, fpk<N>().one);
  fp_load_be<N>(xP, g1);
  fp_load_be<N>(yP, g1 + NB);
  bool valid;
  {
    // QR = Q + R, affine (element_add(QR, Q, p->R), e_param.c:479)
    fp<N> xQ, yQ, l, t, x3, y3;
    fp_load_be<N>(xQ, g2);
    fp_load_be<N>(yQ, g2 + NB);
    valid = (int) e_on_curve<N>(xP, yP) & (int) e_on_curve<N>(xQ, yQ);
    fp_sub<N>(t, K.x2, xQ);
    fp_inv<N>(t, t);
    fp_sub<N>(l, K.y2, yQ);
    fp_mul<N>(l, l, t);
    fp_sqr<N>(x3, l);
    fp_sub<N>(x3, x3, xQ);
    fp_sub<N>(x3, x3, K.x2);
    fp_sub<N>(t, xQ, x3);
    fp_mul<N>(y3, t, l);
    fp_sub<N>(y3, y3, yQ);
    x1 = x3;
    y1 = y3;
  }
  (void) lds_stride;                   // Q + R stays in the lane's private memory (memory operands)
  ejac<N> V;
  fp<N> hot_private[2];
  V.zz = lds_q ? reinterpret_cast<fp<N> *>(lds_q) : &hot_private[0];
  V.t0 = lds_q ? reinterpret_cast<fp<N> *>(lds_q + kWideHotWords / 2) : &hot_private[1];
  V.X = xP; V.Y = yP; V.Z = one; *V.zz = one;
  for (int i = c_e.rbits - 2; i >= 0; i--) {
    e_double_step<N>(n, d, V, x1, y1, K);
    if ((c_e.r[i >> 5] >> (i & 31)) & 1) {
      if constexpr (!kMemOperands<N>) {  // register-resident fields do not keep P live across the loop
        fp_load_be<N>(xP, g1);
        fp_load_be<N>(yP, g1 + NB);
      }
      if (i > 0) {
        e_add_step<N>(n, d, V, xP, yP, x1, y1, K);
      } else {
        // last addition: V = -P, the chord is the vertical through P and V + P = O
        fp_mulx<N>(n, B_SUB, n, n, x1, xP, n, n);
        fp_mulx<N>(d, B_SUB, d, d, K.x2, xP, d, d);
      }
    }
  }
  return valid;
}

// element_pairing (e_pairing) / element_prod_pairing (generic_prod_pairings) for one lane
template <int N>
PBC_DEV void e_prod_pairing_lane(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, int k, uint32_t *lds_q,
                                 int lds_stride) {
  const int L = 2 * fq_bytes<N>();
  fp<N> n, d, out;
  fp_set<N>(n, fpk<N>().one);
  d = n;
  bool valid = e_miller_lane<N>(n, d, g1, g2, lds_q, lds_stride);
  for (int j = 1; j < k; j++) {
    fp<N> nj, dj;
    fp_set<N>(nj, fpk<N>().one);
    dj = nj;
    valid &= e_miller_lane<N>(nj, dj, g1 + (size_t) j * L, g2 + (size_t) j * L, lds_q, lds_stride);
    fp_mul<N>(n, n, nj);
    fp_mul<N>(d, d, dj);
  }
  fp_inv<N>(d, d);
  fp_mul<N>(n, n, d);
  e_pow_win<N>(out, n, c_e.phik, c_e.phikbits, lds_q ? reinterpret_cast<fp<N> *>(lds_q) : nullptr);
  if (!valid) fp_set<N>(out, fpk<N>().one);       // GT identity (pairing_apply, include/pbc_pairing.h:123-130)
  fp_store_be<N>(gt, out);
}

// ---- one-time search of the auxiliary point (single lane) ---------------------------------------
template <int N>
PBC_DEV bool e_sqrt(fp<N> &out, const fp<N> &a, const ERaw &raw) {
  fp<N> one, chk, z, c, x, b, tt, g;
  fp_set<N>(one, fpk<N>().one);
  e_pow<N>(chk, a, raw.half, raw.halfbits);
  if (!fp_eq<N>(chk, one)) return false;
  z = one;
  for (;;) {                           // smallest non-residue 2, 3, ...
    fp_add<N>(z, z, one);
    e_pow<N>(chk, z, raw.half, raw.halfbits);
    if (!fp_eq<N>(chk, one)) break;
  }
  e_pow<N>(c, z, raw.t, raw.tbits);
  e_pow<N>(x, a, raw.t1h, raw.t1hbits);
  e_pow<N>(b, a, raw.t, raw.tbits);
  int m = raw.s;
  while (!fp_eq<N>(b, one)) {
    int i = 0;
    tt = b;
    while (!fp_eq<N>(tt, one)) { fp_sqr<N>(tt, tt); i++; }
    g = c;
    for (int j = 0; j < m - i - 1; j++) fp_sqr<N>(g, g);
    fp_mul<N>(x, x, g);
    fp_sqr<N>(c, g);
    fp_mul<N>(b, b, c);
    m = i;
  }
  out = x;
  return true;
}
// one lane: curve coefficients into Montgomery form and the auxiliary point R (once per parameter set)
template <int N>
PBC_DEV void e_init_lane(EConst *out, const ERaw &raw, const EConst &base) {
  EConst C = base;
  fp<N> r2, t, a, b, one, x, rhs, y;
  fp_set<N>(r2, fpk<N>().r2);
  fp_set<N>(one, fpk<N>().one);
  fp_set<N>(t, raw.a); fp_mul<N>(a, t, r2);
  fp_set<N>(t, raw.b); fp_mul<N>(b, t, r2);
  x = one;
  int xs = 1;
  for (;;) {
    fp_sqr<N>(rhs, x);
    fp_add<N>(rhs, rhs, a);
    fp_mul<N>(rhs, rhs, x);
    fp_add<N>(rhs, rhs, b);
    if (!fp_is0<N>(rhs) && e_sqrt<N>(y, rhs, raw)) break;
    fp_add<N>(x, x, one);
    xs++;
  }
  C.rxs = xs;
  for (int k = 0; k < N; k++) { C.A[k] = a.v[k]; C.B[k] = b.v[k]; C.Rx[k] = x.v[k]; C.Ry[k] = y.v[k]; }
  *out = C;
}

}  // namespace pbc
