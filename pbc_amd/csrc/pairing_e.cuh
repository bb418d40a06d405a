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
//     one (q-1)/r power per lane.
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

template <int N>
struct ejac { fp<N> X, Y, Z, ZZ; };

// n <- n * l(S1) * v(S2),  d <- d * l(S2) * v(S1)   for S1 = Q + R (per lane), S2 = R (uniform)
//   tangent at V scaled by 2 Y Z^3:  l(S) = (Z3 ZZ) ys - 2 Y^2 - M (ZZ xs - X),  M = 3X^2 + a Z^4, Z3 = 2YZ
//   vertical at 2V scaled by Z3^2:   v(S) = Z3^2 xs - X3
template <int N>
PBC_DEV void e_double_step(fp<N> &n, fp<N> &d, ejac<N> &V, const fp<N> &x1, const fp<N> &y1) {
  const fp<N> x2 = ek<N>(c_e.Rx), y2 = ek<N>(c_e.Ry);
  fp<N> XX, YY, M, t0, t1, Z3, W, l1, l2, S;
  fp_sqr<N>(n, n);
  fp_sqr<N>(d, d);
  fp_sqr<N>(XX, V.X);
  fp_sqr<N>(YY, V.Y);
  fp_sqr<N>(t0, V.ZZ);
  fp_mul<N>(t0, t0, ek<N>(c_e.A));
  fp_dbl<N>(M, XX);
  fp_add<N>(M, M, XX);
  fp_add<N>(M, M, t0);
  fp_add<N>(Z3, V.Y, V.Z);
  fp_sqr<N>(Z3, Z3);
  fp_sub<N>(Z3, Z3, YY);
  fp_sub<N>(Z3, Z3, V.ZZ);             // 2YZ
  fp_mul<N>(W, Z3, V.ZZ);
  fp_dbl<N>(t1, YY);                   // 2Y^2
  // l(S1)
  fp_mul<N>(t0, V.ZZ, x1);
  fp_sub<N>(t0, t0, V.X);
  fp_mul<N>(t0, M, t0);
  fp_mul<N>(l1, W, y1);
  fp_sub<N>(l1, l1, t1);
  fp_sub<N>(l1, l1, t0);
  // l(S2)
  fp_mul<N>(t0, V.ZZ, x2);
  fp_sub<N>(t0, t0, V.X);
  fp_mul<N>(t0, M, t0);
  fp_mul<N>(l2, W, y2);
  fp_sub<N>(l2, l2, t1);
  fp_sub<N>(l2, l2, t0);
  fp_mul<N>(n, n, l1);
  fp_mul<N>(d, d, l2);
  // V <- 2V
  fp_mul<N>(S, V.X, YY);
  fp_dbl<N>(S, S);
  fp_dbl<N>(S, S);                     // 4XY^2
  fp_sqr<N>(t0, YY);
  fp_dbl<N>(t0, t0);
  fp_dbl<N>(t0, t0);
  fp_dbl<N>(t0, t0);                   // 8Y^4
  fp_sqr<N>(V.X, M);
  fp_dbl<N>(t1, S);
  fp_sub<N>(V.X, V.X, t1);
  fp_sub<N>(t1, S, V.X);
  fp_mul<N>(t1, M, t1);
  fp_sub<N>(V.Y, t1, t0);
  V.Z = Z3;
  fp_sqr<N>(V.ZZ, Z3);
  // verticals at the new V
  fp_mul<N>(t0, V.ZZ, x2);
  fp_sub<N>(t0, t0, V.X);
  fp_mul<N>(n, n, t0);
  fp_mul<N>(t0, V.ZZ, x1);
  fp_sub<N>(t0, t0, V.X);
  fp_mul<N>(d, d, t0);
}
// chord through V and the affine P scaled by Z3 = Z H:  l(S) = (ys - yP) Z3 - R' (xs - xP),
// H = xP Z^2 - X, R' = yP Z^3 - Y;  then V <- V + P and the verticals at the new V
template <int N>
PBC_DEV void e_add_step(fp<N> &n, fp<N> &d, ejac<N> &V, const fp<N> &xP, const fp<N> &yP, const fp<N> &x1,
                        const fp<N> &y1) {
  const fp<N> x2 = ek<N>(c_e.Rx), y2 = ek<N>(c_e.Ry);
  fp<N> H, Rr, HH, HHH, t0, t1, Z3;
  fp_mul<N>(H, xP, V.ZZ);
  fp_sub<N>(H, H, V.X);
  fp_mul<N>(t0, V.Z, V.ZZ);
  fp_mul<N>(Rr, yP, t0);
  fp_sub<N>(Rr, Rr, V.Y);
  fp_mul<N>(Z3, V.Z, H);
  fp_sub<N>(t0, y1, yP);
  fp_mul<N>(t0, t0, Z3);
  fp_sub<N>(t1, x1, xP);
  fp_mul<N>(t1, t1, Rr);
  fp_sub<N>(t0, t0, t1);
  fp_mul<N>(n, n, t0);
  fp_sub<N>(t0, y2, yP);
  fp_mul<N>(t0, t0, Z3);
  fp_sub<N>(t1, x2, xP);
  fp_mul<N>(t1, t1, Rr);
  fp_sub<N>(t0, t0, t1);
  fp_mul<N>(d, d, t0);
  fp_sqr<N>(HH, H);
  fp_mul<N>(HHH, HH, H);
  fp_mul<N>(t0, V.X, HH);
  fp_sqr<N>(t1, Rr);
  fp_sub<N>(t1, t1, HHH);
  fp_sub<N>(t1, t1, t0);
  fp_sub<N>(t1, t1, t0);
  fp_sub<N>(t0, t0, t1);
  fp_mul<N>(t0, Rr, t0);
  fp_mul<N>(HHH, V.Y, HHH);
  fp_sub<N>(V.Y, t0, HHH);
  V.X = t1;
  V.Z = Z3;
  fp_sqr<N>(V.ZZ, Z3);
  fp_mul<N>(t0, V.ZZ, x2);
  fp_sub<N>(t0, t0, V.X);
  fp_mul<N>(n, n, t0);
  fp_mul<N>(t0, V.ZZ, x1);
  fp_sub<N>(t0, t0, V.X);
  fp_mul<N>(d, d, t0);
}

// numerator and denominator of f_{r,P}(Q+R) / f_{r,P}(R) for one lane (n = d = 1 on entry).
// G1, G2 bytes: x||y.  Returns false when an input deserialises to O (curve_from_bytes,
// ecc/curve.c:609-623).
template <int N>
PBC_DEV bool e_miller_lane(fp<N> &n, fp<N> &d, const uint8_t *g1, const uint8_t *g2, uint32_t *lds_q,
                           int lds_stride) {
  const int NB = fq_bytes<N>();
  fp<N> one, xP, yP, x1, y1;
  fp_set<N>(one, fpk<N>().one);
  fp_load_be<N>(xP, g1);
  fp_load_be<N>(yP, g1 + NB);
  bool valid;
  {
    // QR = Q + R, affine (element_add(QR, Q, p->R), e_param.c:479)
    const fp<N> xR = ek<N>(c_e.Rx), yR = ek<N>(c_e.Ry);
    fp<N> xQ, yQ, l, t, x3, y3;
    fp_load_be<N>(xQ, g2);
    fp_load_be<N>(yQ, g2 + NB);
    valid = (int) e_on_curve<N>(xP, yP) & (int) e_on_curve<N>(xQ, yQ);
    fp_sub<N>(t, xR, xQ);
    fp_inv<N>(t, t);
    fp_sub<N>(l, yR, yQ);
    fp_mul<N>(l, l, t);
    fp_sqr<N>(x3, l);
    fp_sub<N>(x3, x3, xQ);
    fp_sub<N>(x3, x3, xR);
    fp_sub<N>(t, xQ, x3);
    fp_mul<N>(y3, t, l);
    fp_sub<N>(y3, y3, yQ);
    x1 = x3;
    y1 = y3;
  }
  (void) lds_q; (void) lds_stride;     // Q + R stays in the lane's private memory (memory operands)
  ejac<N> V;
  V.X = xP; V.Y = yP; V.Z = one; V.ZZ = one;
  for (int i = c_e.rbits - 2; i >= 0; i--) {
    e_double_step<N>(n, d, V, x1, y1);
    if ((c_e.r[i >> 5] >> (i & 31)) & 1) {
      fp_load_be<N>(xP, g1);
      fp_load_be<N>(yP, g1 + NB);
      if (i > 0) {
        e_add_step<N>(n, d, V, xP, yP, x1, y1);
      } else {
        // last addition: V = -P, the chord is the vertical through P and V + P = O
        fp<N> t;
        fp_sub<N>(t, x1, xP);
        fp_mul<N>(n, n, t);
        fp_sub<N>(t, ek<N>(c_e.Rx), xP);
        fp_mul<N>(d, d, t);
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
  e_pow<N>(out, n, c_e.phik, c_e.phikbits);
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
  for (;;) {
    fp_sqr<N>(rhs, x);
    fp_add<N>(rhs, rhs, a);
    fp_mul<N>(rhs, rhs, x);
    fp_add<N>(rhs, rhs, b);
    if (!fp_is0<N>(rhs) && e_sqrt<N>(y, rhs, raw)) break;
    fp_add<N>(x, x, one);
  }
  for (int k = 0; k < N; k++) { C.A[k] = a.v[k]; C.B[k] = b.v[k]; C.Rx[k] = x.v[k]; C.Ry[k] = y.v[k]; }
  *out = C;
}

}  // namespace pbc
