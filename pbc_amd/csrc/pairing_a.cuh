// pairing_a.cuh -- Type A (supersingular y^2 = x^3 + x over F_q, k = 2) reduced Tate
// pairing, one pairing per lane.
//
// Computes the same GT value as the reference's default Type-A map a_pairing_proj
// (ecc/a_param.c:1053-1198) followed by a_tateexp (:285-303), but is re-derived for a GPU:
//   * the Miller loop never leaves Jacobian coordinates: the two point_to_affine()
//     inversions of the reference (:1073-1080) are gone, the one addition is a mixed
//     Jacobian+affine step (the reference itself notes this is possible, :1068-1069).
//     All scale factors introduced are in F_q^* and vanish in the final exponentiation.
//   * tangent line and doubling share X^2, Y^2, Z^4 (10 M + 8 S per step vs 23 M).
//   * the final exponentiation needs ONE inversion instead of two: with f = a+bi,
//     N = a^2+b^2, A = a^2-b^2, B = -2ab we have f^(q-1) = (A+Bi)/N, and the closing
//     division of lucas_odd (:272-281) by P^2-4 = -4 (B/N)^2 folds into N/B; both 1/N and
//     1/B come from one inversion of N*B.
// Field values are exact residues, so any algebraically equal evaluation order yields
// bit-identical canonical bytes.
#pragma once
#include "fp.cuh"

#ifndef PBC_AP_FAIR_BIT
#define PBC_AP_FAIR_BIT 23                // priority slices of the word-form type a kernels (fp.cuh pbc_fair_tick)
#endif
namespace pbc {

// F_q^2 = F_q[i]/(i^2+1)   (arith/fieldquadratic.c fi_*; q = 3 mod 4)
template <int N>
struct fp2 {
  fp<N> x, y;
};

// fi_mul (fieldquadratic.c:425-457): Karatsuba, 3 M
template <int N>
PBC_DEV void fi_mul(fp2<N> &r, const fp2<N> &a, const fp2<N> &b) {
  fp<N> e0, e1, e2;
  fp_add<N>(e0, a.x, a.y);
  fp_add<N>(e1, b.x, b.y);
  fp_mul<N>(e2, e0, e1);
  fp_mul<N>(e0, a.x, b.x);
  fp_mul<N>(e1, a.y, b.y);
  fp_sub<N>(e2, e2, e0);
  fp_sub<N>(r.x, e0, e1);
  fp_sub<N>(r.y, e2, e1);
}
// fi_square (fieldquadratic.c:459-477): (x+y)(x-y) + 2xy i, 2 M
template <int N>
PBC_DEV void fi_sqr(fp2<N> &r, const fp2<N> &a) {
  fp<N> e0, e1;
  fp_add<N>(e0, a.x, a.y);
  fp_sub<N>(e1, a.x, a.y);
  fp_mul<N>(e0, e0, e1);
  fp_mul<N>(e1, a.x, a.y);
  fp_dbl<N>(r.y, e1);
  r.x = e0;
}

// The same two routines on fused products (fp.cuh "Fused products"; the wide fields): no sum or difference goes through
// memory on its own.  r must not overlap a or b.
// (R, A, B: fp2<N> or the view fp2v<N> of two separately placed elements)
template <int N>
struct fp2v {
  fp<N> &x, &y;
};
template <int N, class R, class A, class B>
PBC_DEV void fi_mul_x(R &r, const A &a, const B &b) {
  using namespace fx;
  // two two-product sums with one reduction each (fp_sopx): the multiply-adds of Karatsuba's three products, two calls
  // instead of three and 10 element moves instead of 14
#if PBC_WIDE_NO_SOP                    // A/B: Karatsuba, three products with a reduction each
  fp<N> e1;
  fp_mulx<N>(e1, a.y, b.y);
  fp_mulx<N>(r.x, C1_SUB, a.x, a.x, b.x, b.x, e1, e1);
  fp_mulx<N>(r.y, A_ADD | B_ADD | C1_SUB | C2_SUB | c2_sh(1), a.x, a.y, b.x, b.y, r.x, e1);
#else
  fp_sopx<N>(r.x, NEG2, a.x, b.x, a.y, b.y, b.y, a.x, a.x);                                // a.x b.x - a.y b.y
  fp_sopx<N>(r.y, 0, a.x, b.y, a.y, b.x, b.x, a.x, a.x);                                   // a.x b.y + a.y b.x
#endif
}
template <int N, class R, class A>
PBC_DEV void fi_sqr_x(R &r, const A &a) {
  using namespace fx;
  fp_mulx<N>(r.x, A_ADD | B_SUB, a.x, a.y, a.x, a.y, a.x, a.x);
  fp_mulx<N>(r.y, dbl(1), a.x, a.x, a.y, a.y, a.x, a.x);
}

struct AConst {          // a_pairing_data (ecc/a_param.c:30-34) + phikonr = h (:1458)
  uint32_t h[34];        // cofactor h = (q+1)/r, little-endian words (type a1: l, a_param.c:2242-2244)
  int hbits;
  int exp2, exp1, sign1; // r = 2^exp2 + sign1 2^exp1 + sign0 (sign0 unused by the map)
  uint32_t sqrt_e[34];   // (q + 1)/4: square roots in F_q for q = 3 mod 4 (element_from_hash)
  int sqrt_bits;
  uint32_t r[34], rm[34]; // type a1 (and type a outside the 512-bit fast path): Miller loop digits, NAF of n >> 1 (+1 / -1)
  int rbits;
  // limb-form kernel (pairing_al.cuh): multiples c q of the modulus in 29-bit limbs, "borrowed" so that limb i
  // dominates limb i of any subtrahend with limbs <= D (2^29 - 1) and value < c q:
  //     limb_0 + D 2^29,   limb_i + D 2^29 - D  (0 < i < 17),   limb_17 - D;      (c, D) = (2,1) (4,2) (8,4) (12,2) (16,2)
  uint32_t ksub[5][18];
};
static_assert(sizeof(AConst) <= KOFF_XS - KOFF_TYPE, "constant block layout");
#define c_a (pbc::kconst<pbc::AConst, pbc::KOFF_TYPE>())
constexpr int AW_AUX_HEAD = 4;   // the table of the wave kernels of type a1 / generic type a (pairing_aw.cuh AG<N>, host_params.h ag_aux_build): words before its five constants; [0] = LEFF

// A first argument with y = 0 is a point of order 2 ((0, 0): the zero-filled record).  Its tangent is vertical and the
// reference divides by zero (point_to_affine / the slope 1/2y); the pairing value is 1 since 2 is coprime to the group
// order, which is what this engine returns by treating such a P like O (include/pbc_hip.h, "zero-filled records").
template <int N>
PBC_DEV bool a_first_arg_ok(const fp<N> &x, const fp<N> &y);
// curve_is_valid_point (ecc/curve.c:57-77) for y^2 = x^3 + x
template <int N>
PBC_DEV bool a_on_curve(const fp<N> &x, const fp<N> &y) {
  fp<N> t0, t1, one;
  fp_set<N>(one, fpk<N>().one);
  fp_sqr<N>(t0, x);
  fp_add<N>(t0, t0, one);
  fp_mul<N>(t0, t0, x);
  fp_sqr<N>(t1, y);
  return fp_eq<N>(t0, t1);
}

template <int N>
PBC_DEV bool a_first_arg_ok(const fp<N> &x, const fp<N> &y) { return (int) a_on_curve<N>(x, y) & (int) !fp_is0<N>(y); }

template <int N>
struct jac {
  fp<N> X, Y, Z, ZZ;     // x = X/Z^2, y = Y/Z^3, ZZ = Z^2 cached
};

// One doubling step of the Miller loop: f <- f^2 * l_{V,V}(phi(Q)), V <- 2V.
// phi(x,y) = (-x, iy) is the distortion map (a_miller_evalfn, a_param.c:306-315).
// Line (scaled by 2 Y Z^3 in F_q^*):  re = M (ZZ Qx + X) - 2 Y^2,  im = (2YZ) ZZ Qy,
// with M = 3X^2 + Z^4.
// SQR = false leaves the squaring of f to the caller (the product kernel squares its shared accumulator once per
// iteration for all terms, as a_pairings_affine does, ecc/a_param.c:1338-1344).
// `hot` (the 33-word fields on the device): 72 words of LDS owned by the lane, for the two temporaries of the step that move
// most often (Y^2 and X^2: 9 of its 66 element moves; 16-byte aligned at words 0 and 36)
template <int N, bool SQR = true>
PBC_DEV void a_double_step(fp2<N> &f, jac<N> &V, const fp<N> &Qx, const fp<N> &Qy, uint32_t *hot = nullptr) {
  // Ordered for short live ranges (the register budget is 256/lane at 2 waves per SIMD):
  // line first, f <- f^2 l as soon as the line exists, the rest of the doubling last.
  // Two products are traded for squarings (a dedicated squaring costs 0.78 of a product):
  //   2YZ = (Y+Z)^2 - Y^2 - Z^2,   4XY^2 = 2((X+Y^2)^2 - X^2 - Y^4)      -> 10 M + 8 S per step
  if constexpr (kMemOperands<N> && SQR) {
    // the wide fields: the same 10 M + 8 S as 18 fused products, f through a second buffer instead of in place
    using namespace fx;
    fp<N> hot_private[2], M, t0, S, Y4;
    fp<N> &YY = hot ? *reinterpret_cast<fp<N> *>(hot) : hot_private[0];
    fp<N> &XX = hot ? *reinterpret_cast<fp<N> *>(hot + 36) : hot_private[1];
    fp2<N> l, g;
    fi_sqr_x<N>(g, f);
    fp_sqrx<N>(XX, V.X);
    fp_sqrx<N>(M, C1_ADD | c1_sh(1) | C2_ADD, V.ZZ, V.ZZ, XX, XX);                       // M = Z^4 + 2X^2 + X^2
    fp_sqrx<N>(YY, V.Y);
    fp_mulx<N>(t0, C1_ADD, V.ZZ, V.ZZ, Qx, Qx, V.X, V.X);
    fp_mulx<N>(l.x, C1_SUB | c1_sh(1), M, M, t0, t0, YY, YY);                            // re = M (ZZ Qx + X) - 2Y^2
    fp_sqrx<N>(V.Z, A_ADD | C1_SUB | C2_SUB, V.Y, V.Z, YY, V.ZZ);                        // Z3 = 2YZ, in place (Y, Z dead)
    fp_mulx<N>(t0, V.Z, V.ZZ);
    fp_mulx<N>(l.y, t0, Qy);                                                              // im = Z3 ZZ Qy
    fi_mul_x<N>(f, g, l);
    fp_sqrx<N>(Y4, YY);
    fp_sqrx<N>(S, A_ADD | C1_SUB | C2_SUB | dbl(1), V.X, YY, XX, Y4);                     // S = 4XY^2
    fp_sqrx<N>(V.X, C1_SUB | c1_sh(1), M, M, S, S);                                      // X3 = M^2 - 2S
    fp_mulx<N>(V.Y, B_SUB | C1_SUB | c1_sh(3), M, M, S, V.X, Y4, Y4);                    // Y3 = M(S - X3) - 8Y^4
    fp_sqrx<N>(V.ZZ, V.Z);
    return;
  }
  fp<N> XX, YY, M, t0, t1, S, Z3, Y4;
  fp2<N> l;
  if constexpr (SQR) fi_sqr<N>(f, f);
  fp_sqr<N>(XX, V.X);
  fp_sqr<N>(t0, V.ZZ);                 // Z^4
  fp_dbl<N>(M, XX);
  fp_add<N>(M, M, XX);
  fp_add<N>(M, M, t0);                 // M = 3X^2 + a Z^4, a = 1
  fp_sqr<N>(YY, V.Y);
  fp_mul<N>(t0, V.ZZ, Qx);
  fp_add<N>(t0, t0, V.X);
  fp_mul<N>(l.x, M, t0);
  fp_dbl<N>(t1, YY);
  fp_sub<N>(l.x, l.x, t1);             // re = M (ZZ Qx + X) - 2Y^2
  fp_add<N>(Z3, V.Y, V.Z);
  fp_sqr<N>(Z3, Z3);
  fp_sub<N>(Z3, Z3, YY);
  fp_sub<N>(Z3, Z3, V.ZZ);             // Z3 = 2YZ            (Y, Z dead)
  fp_mul<N>(t1, Z3, V.ZZ);             //                      (ZZ dead)
  fp_mul<N>(l.y, t1, Qy);              // im = Z3 ZZ Qy
  fi_mul<N>(f, f, l);                  //                      (l dead)
  fp_sqr<N>(Y4, YY);                   // Y^4
  fp_add<N>(S, V.X, YY);
  fp_sqr<N>(S, S);
  fp_sub<N>(S, S, XX);
  fp_sub<N>(S, S, Y4);
  fp_dbl<N>(S, S);                     // S = 4XY^2            (X, YY dead)
  fp_dbl<N>(t0, Y4);
  fp_dbl<N>(t0, t0);
  fp_dbl<N>(t0, t0);                   // 8Y^4
  fp_sqr<N>(V.X, M);
  fp_dbl<N>(t1, S);
  fp_sub<N>(V.X, V.X, t1);             // X3 = M^2 - 2S
  fp_sub<N>(t1, S, V.X);
  fp_mul<N>(t1, M, t1);
  fp_sub<N>(V.Y, t1, t0);              // Y3 = M(S - X3) - 8Y^4
  V.Z = Z3;
  fp_sqr<N>(V.ZZ, Z3);
}

// Mixed addition step of the Miller loop: f <- f * l_{V,P}(phi(Q)), V <- V + P, with
// P = (x2, y2) affine.  H = x2 Z^2 - X, R = y2 Z^3 - Y, Z3 = Z H; the chord has slope
// R/Z3, so scaled by Z3 in F_q^*:   re = R (Qx + x2) - Z3 y2,   im = Z3 Qy.
// (affine original: compute_abc_line a_param.c:114-130 + a_miller_evalfn :306-315)
template <int N>
PBC_DEV void a_add_step(fp2<N> &f, jac<N> &V, const fp<N> &x2, const fp<N> &y2, const fp<N> &Qx,
                        const fp<N> &Qy) {
  if constexpr (kMemOperands<N>) {
    using namespace fx;
    fp<N> H, R, HH, HHH, t0;
    fp2<N> l, g;
    fp_mulx<N>(H, C1_SUB, x2, x2, V.ZZ, V.ZZ, V.X, V.X);
    fp_mulx<N>(t0, V.Z, V.ZZ);
    fp_mulx<N>(R, C1_SUB, y2, y2, t0, t0, V.Y, V.Y);
    fp_mulx<N>(V.Z, V.Z, H);                                                              // Z3, in place
    fp_mulx<N>(t0, V.Z, y2);
    fp_mulx<N>(l.x, B_ADD | C1_SUB, R, R, Qx, x2, t0, t0);                                // re = R (Qx + x2) - Z3 y2
    fp_mulx<N>(l.y, V.Z, Qy);
    fp_sqrx<N>(HH, H);
    fp_mulx<N>(HHH, HH, H);
    fp_mulx<N>(t0, V.X, HH);                                                              // X1 H^2
    fp_sqrx<N>(V.X, C1_SUB | C2_SUB | c2_sh(1), R, R, HHH, t0);                           // X3 = R^2 - H^3 - 2 X1 H^2
    fp_mulx<N>(HHH, V.Y, HHH);
    fp_mulx<N>(V.Y, B_SUB | C1_SUB, R, R, t0, V.X, HHH, HHH);                             // Y3 = R (X1 H^2 - X3) - Y1 H^3
    fp_sqrx<N>(V.ZZ, V.Z);
    fi_mul_x<N>(g, f, l);
    f = g;
    return;
  }
  fp<N> H, R, HH, HHH, t0, t1, Z3;
  fp2<N> l;
  fp_mul<N>(H, x2, V.ZZ);
  fp_sub<N>(H, H, V.X);
  fp_mul<N>(t0, V.Z, V.ZZ);
  fp_mul<N>(R, y2, t0);
  fp_sub<N>(R, R, V.Y);
  fp_mul<N>(Z3, V.Z, H);
  fp_add<N>(t0, Qx, x2);
  fp_mul<N>(l.x, R, t0);
  fp_mul<N>(t0, Z3, y2);
  fp_sub<N>(l.x, l.x, t0);
  fp_mul<N>(l.y, Z3, Qy);
  fp_sqr<N>(HH, H);
  fp_mul<N>(HHH, HH, H);
  fp_mul<N>(t0, V.X, HH);              // X1 H^2
  fp_sqr<N>(t1, R);
  fp_sub<N>(t1, t1, HHH);
  fp_sub<N>(t1, t1, t0);
  fp_sub<N>(t1, t1, t0);               // X3 = R^2 - H^3 - 2 X1 H^2
  fp_sub<N>(t0, t0, t1);
  fp_mul<N>(t0, R, t0);
  fp_mul<N>(HHH, V.Y, HHH);
  fp_sub<N>(V.Y, t0, HHH);             // Y3 = R (X1 H^2 - X3) - Y1 H^3
  V.X = t1;
  V.Z = Z3;
  fp_sqr<N>(V.ZZ, Z3);
  fi_mul<N>(f, f, l);
}

// f^((q^2-1)/r) = (f^(q-1))^h via the Lucas V-sequence (a_tateexp + lucas_odd,
// a_param.c:226-303), single inversion.
template <int N>
PBC_DEV void a_final_exp(fp2<N> &out, const fp2<N> &f) {
  fp<N> a2, b2, Nn, A, B, t, g0, P, v0, v1, two, w;
  fp_sqr<N>(a2, f.x);
  fp_sqr<N>(b2, f.y);
  fp_add<N>(Nn, a2, b2);
  fp_sub<N>(A, a2, b2);
  fp_mul<N>(B, f.x, f.y);
  fp_dbl<N>(B, B);
  fp_neg<N>(B, B);                     // f^(q-1) = conj(f)^2 / N = (A + B i) / N
  // (B = 0: f^(q-1) = +-1.  Invert N * 1 instead; the imaginary part below is then (2 V_{h+1} - P V_h) N / 4 = 0
  // exactly, since P = +-2 gives V_n = 2 (+-1)^n.)
  fp_set<N>(t, fpk<N>().one);
  fp_cmov<N>(B, t, fp_is0<N>(B));
  fp_mul<N>(t, Nn, B);
  fp_inv<N>(t, t);                     // 1/(N B)
  fp_mul<N>(w, t, B);                  // 1/N
  fp_mul<N>(g0, A, w);                 // Re f^(q-1)
  fp_mul<N>(t, t, Nn);                 // 1/B
  fp_mul<N>(w, t, Nn);                 // N/B = 1/Im f^(q-1)
  fp_dbl<N>(P, g0);                    // t1 = 2 in0
  fp_set<N>(two, fpk<N>().one);
  fp_dbl<N>(two, two);                 // t0 = 2
  v0 = two;
  v1 = P;
  for (int j = c_a.hbits - 1; j >= 0; j--) {
    if ((j & 15) == 0) pbc_fair_tick<PBC_AP_FAIR_BIT>();
    bool bit = j ? ((c_a.h[j >> 5] >> (j & 31)) & 1) : false;   // j == 0 runs the 0-branch (:243-249)
    fp<N> m, s;
    fp_mul<N>(m, v0, v1);
    fp_sub<N>(m, m, P);
    if (bit) {
      fp_sqr<N>(s, v1);
      fp_sub<N>(v1, s, two);
      v0 = m;
    } else {
      fp_sqr<N>(s, v0);
      fp_sub<N>(v0, s, two);
      v1 = m;
    }
  }
  // out.y = (2 v1 - P v0) / (P^2 - 4) * Im(f^(q-1)) = -(2 v1 - P v0) (N/B) / 4
  fp_mul<N>(t, v0, P);
  fp_dbl<N>(v1, v1);
  fp_sub<N>(v1, v1, t);
  fp_mul<N>(v1, v1, w);
  fp_halve<N>(v1, v1);
  fp_halve<N>(v1, v1);
  fp_neg<N>(out.y, v1);
  fp_halve<N>(out.x, v0);
}

template <int N>
PBC_DEV void a_store_gt(uint8_t *gt, fp2<N> &out, bool valid) {
  if (!valid) {                        // GT identity (pairing_apply, include/pbc_pairing.h:123-130)
    fp_set<N>(out.x, fpk<N>().one);
#pragma unroll
    for (int k = 0; k < N; k++) out.y.v[k] = 0;
  }
  fp_store_be<N>(gt, out.x);
  fp_store_be<N>(gt + fq_bytes<N>(), out.y);
}

// ---- preprocessed pairings: pairing_pp_init / pairing_pp_apply (include/pbc_pairing.h:54-89) ----
// a_pairing_pp_init (a_param.c:149-220) stores the line coefficients of every Miller step for a
// fixed first argument; a_pairing_pp_apply (:317-360) then needs no point arithmetic.  Here the
// table holds, per step, (cA, cB, cC) with  line(Q) = (cA Qx + cC) + i (cB Qy)  in the same
// projective scaling the full kernel uses: doubling  cA = M ZZ, cB = Z3 ZZ, cC = M X - 2Y^2;
// addition  cA = R, cB = Z3, cC = R x2 - Z3 y2.  Layout: [exp2 + 1][3][N] words (Montgomery form),
// entry exp2 is the addition step.  The table is wave-uniform data for the apply kernel.
template <int N>
PBC_DEV void a_pp_store(uint32_t *tab, int idx, const fp<N> &cA, const fp<N> &cB, const fp<N> &cC) {
#pragma unroll
  for (int k = 0; k < N; k++) {
    tab[(idx * 3 + 0) * N + k] = cA.v[k];
    tab[(idx * 3 + 1) * N + k] = cB.v[k];
    tab[(idx * 3 + 2) * N + k] = cC.v[k];
  }
}
// line coefficients of the tangent at V, then V <- 2V
template <int N>
PBC_DEV void a_pp_dbl(jac<N> &V, fp<N> &cA, fp<N> &cB, fp<N> &cC) {
  fp<N> XX, YY, M, t0, t1, S, Z3;
  fp_sqr<N>(XX, V.X);
  fp_sqr<N>(YY, V.Y);
  fp_sqr<N>(t0, V.ZZ);
  fp_dbl<N>(M, XX);
  fp_add<N>(M, M, XX);
  fp_add<N>(M, M, t0);
  fp_mul<N>(cA, M, V.ZZ);
  fp_mul<N>(Z3, V.Y, V.Z);
  fp_dbl<N>(Z3, Z3);
  fp_mul<N>(cB, Z3, V.ZZ);
  fp_mul<N>(cC, M, V.X);
  fp_dbl<N>(t1, YY);
  fp_sub<N>(cC, cC, t1);
  fp_mul<N>(S, V.X, YY);
  fp_dbl<N>(S, S);
  fp_dbl<N>(S, S);
  fp_sqr<N>(t0, YY);
  fp_dbl<N>(t0, t0);
  fp_dbl<N>(t0, t0);
  fp_dbl<N>(t0, t0);
  fp_sqr<N>(V.X, M);
  fp_dbl<N>(t1, S);
  fp_sub<N>(V.X, V.X, t1);
  fp_sub<N>(t1, S, V.X);
  fp_mul<N>(t1, M, t1);
  fp_sub<N>(V.Y, t1, t0);
  V.Z = Z3;
  fp_sqr<N>(V.ZZ, Z3);
}
// line coefficients of the chord through V and the affine (x2, y2), then V <- V + (x2, y2)
template <int N>
PBC_DEV void a_pp_add(jac<N> &V, const fp<N> &x2, const fp<N> &y2, fp<N> &cA, fp<N> &cB, fp<N> &cC) {
  fp<N> H, R, HH, HHH, t0, t1, Z3;
  fp_mul<N>(H, x2, V.ZZ);
  fp_sub<N>(H, H, V.X);
  fp_mul<N>(t0, V.Z, V.ZZ);
  fp_mul<N>(R, y2, t0);
  fp_sub<N>(R, R, V.Y);
  fp_mul<N>(Z3, V.Z, H);
  fp_mul<N>(cC, R, x2);
  fp_mul<N>(t0, Z3, y2);
  fp_sub<N>(cC, cC, t0);
  cA = R;
  cB = Z3;
  fp_sqr<N>(HH, H);
  fp_mul<N>(HHH, HH, H);
  fp_mul<N>(t0, V.X, HH);
  fp_sqr<N>(t1, R);
  fp_sub<N>(t1, t1, HHH);
  fp_sub<N>(t1, t1, t0);
  fp_sub<N>(t1, t1, t0);
  fp_sub<N>(t0, t0, t1);
  fp_mul<N>(t0, R, t0);
  fp_mul<N>(HHH, V.Y, HHH);
  fp_sub<N>(V.Y, t0, HHH);
  V.X = t1;
  V.Z = Z3;
  fp_sqr<N>(V.ZZ, Z3);
}
// one lane: returns false when g1 deserialises to O (then every pp_apply result is 1,
// pairing_pp_init include/pbc_pairing.h:54-61)
template <int N>
PBC_DEV bool a_pp_init_lane(uint32_t *tab, const uint8_t *g1) {
  const int NB = fq_bytes<N>();
  fp<N> one, x2, y2, cA, cB, cC;
  jac<N> V;
  fp_set<N>(one, fpk<N>().one);
  fp_load_be<N>(x2, g1);
  fp_load_be<N>(y2, g1 + NB);
  bool valid = a_first_arg_ok<N>(x2, y2);
  V.X = x2; V.Y = y2; V.Z = one; V.ZZ = one;
  int slot = 0;
  for (int i = c_a.exp2 - 1; i >= 0; i--, slot++) {
    a_pp_dbl<N>(V, cA, cB, cC);
    a_pp_store<N>(tab, slot, cA, cB, cC);
    if (i == c_a.exp1) {
      fp<N> yy = y2;
      if (c_a.sign1 < 0) fp_neg<N>(yy, yy);
      a_pp_add<N>(V, x2, yy, cA, cB, cC);
      a_pp_store<N>(tab, c_a.exp2, cA, cB, cC);
    }
  }
  return valid;
}
// r <- a * line(Q) on fused products (the wide fields; r and a distinct): cA, cB, cC are read where they lie
template <int N, class R, class A>
PBC_DEV void a_pp_line_x(R &r, const A &a, const uint32_t *tab, int idx, const fp<N> &Qx, const fp<N> &Qy) {
  using namespace fx;
  const fp<N> *e = reinterpret_cast<const fp<N> *>(tab + (size_t) idx * 3 * N);
  fp2<N> l;
  fp_mulx<N>(l.x, C1_ADD, e[0], e[0], Qx, Qx, e[2], e[2]);
  fp_mulx<N>(l.y, e[1], Qy);
  fi_mul_x<N>(r, a, l);
}
// f <- f * line(Q) for table entry idx (used by the type a1 apply kernel; a.param's runs in limb form, pairing_al.cuh)
template <int N>
PBC_DEV void a_pp_line(fp2<N> &f, const uint32_t *tab, int idx, const fp<N> &Qx, const fp<N> &Qy) {
  fp<N> cA, cB, cC;
  fp2<N> l;
#pragma unroll
  for (int k = 0; k < N; k++) {
    cA.v[k] = tab[(idx * 3 + 0) * N + k];
    cB.v[k] = tab[(idx * 3 + 1) * N + k];
    cC.v[k] = tab[(idx * 3 + 2) * N + k];
  }
  fp_mul<N>(l.x, cA, Qx);
  fp_add<N>(l.x, l.x, cC);
  fp_mul<N>(l.y, cB, Qy);
  fi_mul<N>(f, f, l);
}
// (element_pairing and pairing_pp_apply for the 512-bit field of a.param run in limb form: pairing_al.cuh)

// element_prod_pairing for one lane: prod_j e(P_j, Q_j) (a_pairings_affine, a_param.c:1283-1383).  As in the
// reference, ONE accumulator serves all k terms: it is squared once per Miller iteration (:1338-1344,
// do_tangents :1296-1306) and multiplied by the k line values, and the final exponentiation runs once.  The k points
// V_j advance in lockstep (element_multi_double, ecc/curve.c:210-281 -- here Jacobian, no inversions); their state and
// the Montgomery forms of the Q_j do not fit registers or LDS (96 words per term), so they live in a global
// workspace owned by the pairing object, laid out so that every access of a wave is one contiguous kilobyte:
//     ws[((block k + term) 24 + 4 element + quad) 128 + lane]  (uint4),   elements X, Y, Z, ZZ, Qx, Qy.
// Any identity input forces the product to 1 (element_prod_pairing, include/pbc_pairing.h:161-168).
template <int N>
PBC_DEV void a_ws_put(uint4 *ws, int e, const fp<N> &a) {
  static_assert(N % 4 == 0, "workspace records are whole uint4s");
#pragma unroll
  for (int q = 0; q < N / 4; q++) ws[(e * (N / 4) + q) * 128] = make_uint4(a.v[4 * q], a.v[4 * q + 1], a.v[4 * q + 2], a.v[4 * q + 3]);
}
template <int N>
PBC_DEV void a_ws_get(fp<N> &a, const uint4 *ws, int e) {
#pragma unroll
  for (int q = 0; q < N / 4; q++) {
    uint4 t = ws[(e * (N / 4) + q) * 128];
    a.v[4 * q] = t.x; a.v[4 * q + 1] = t.y; a.v[4 * q + 2] = t.z; a.v[4 * q + 3] = t.w;
  }
}
// The shared accumulator f waits in LDS (limb-major, [2 N][lanes]) while a term's point arithmetic runs and is touched
// only by the two f <- f * l updates: 32 fewer live registers in the term loop.
template <int N>
PBC_DEV void a_lds_get(fp2<N> &f, const uint32_t *lds, int stride) {
#pragma unroll
  for (int i = 0; i < N; i++) { f.x.v[i] = lds[i * stride]; f.y.v[i] = lds[(N + i) * stride]; }
}
template <int N>
PBC_DEV void a_lds_put(uint32_t *lds, int stride, const fp2<N> &f) {
#pragma unroll
  for (int i = 0; i < N; i++) { lds[i * stride] = f.x.v[i]; lds[(N + i) * stride] = f.y.v[i]; }
}
// One doubling step of a product term: the line value goes into the LDS accumulator, V <- 2V.  Same formulas as
// a_double_step; split in two so that the caller can start fetching the next term's state between the halves.
template <int N>
PBC_DEV void a_prod_double_head(jac<N> &V, fp<N> &M, fp<N> &XX, fp<N> &YY, fp<N> &Z3, const fp<N> &Qx, const fp<N> &Qy,
                                uint32_t *lds_f, int stride) {
  fp<N> t0, t1;
  fp2<N> l, f;
  fp_sqr<N>(XX, V.X);
  fp_sqr<N>(t0, V.ZZ);                 // Z^4
  fp_dbl<N>(M, XX);
  fp_add<N>(M, M, XX);
  fp_add<N>(M, M, t0);                 // M = 3X^2 + Z^4
  fp_sqr<N>(YY, V.Y);
  fp_mul<N>(t0, V.ZZ, Qx);
  fp_add<N>(t0, t0, V.X);
  fp_mul<N>(l.x, M, t0);
  fp_dbl<N>(t1, YY);
  fp_sub<N>(l.x, l.x, t1);             // re = M (ZZ Qx + X) - 2Y^2
  fp_add<N>(Z3, V.Y, V.Z);
  fp_sqr<N>(Z3, Z3);
  fp_sub<N>(Z3, Z3, YY);
  fp_sub<N>(Z3, Z3, V.ZZ);             // Z3 = 2YZ
  fp_mul<N>(t1, Z3, V.ZZ);
  fp_mul<N>(l.y, t1, Qy);              // im = Z3 ZZ Qy
  a_lds_get<N>(f, lds_f, stride);
  fi_mul<N>(f, f, l);
  a_lds_put<N>(lds_f, stride, f);
}
template <int N>
PBC_DEV void a_prod_double_tail(jac<N> &V, const fp<N> &M, const fp<N> &XX, const fp<N> &YY, const fp<N> &Z3) {
  fp<N> S, Y4, t0, t1;
  fp_sqr<N>(Y4, YY);
  fp_add<N>(S, V.X, YY);
  fp_sqr<N>(S, S);
  fp_sub<N>(S, S, XX);
  fp_sub<N>(S, S, Y4);
  fp_dbl<N>(S, S);                     // S = 4XY^2
  fp_dbl<N>(t0, Y4);
  fp_dbl<N>(t0, t0);
  fp_dbl<N>(t0, t0);                   // 8Y^4
  fp_sqr<N>(V.X, M);
  fp_dbl<N>(t1, S);
  fp_sub<N>(V.X, V.X, t1);             // X3 = M^2 - 2S
  fp_sub<N>(t1, S, V.X);
  fp_mul<N>(t1, M, t1);
  fp_sub<N>(V.Y, t1, t0);              // Y3 = M(S - X3) - 8Y^4
  V.Z = Z3;
  fp_sqr<N>(V.ZZ, Z3);
}
template <int N>
PBC_DEV void a_prod_pairing_lane(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, int k, uint4 *ws, uint32_t *lds_f, int stride) {
  constexpr int L = 8 * N, NB = 4 * N, REC = 6 * (N / 4) * 128;      // bytes per record / coordinate, uint4s per term
  fp<N> one;
  fp_set<N>(one, fpk<N>().one);
  bool valid = true;
  for (int j = 0; j < k; j++) {
    fp<N> x, y, Qx, Qy;
    fp_load_be<N>(x, g1 + (size_t) j * L);
    fp_load_be<N>(y, g1 + (size_t) j * L + NB);
    fp_load_be<N>(Qx, g2 + (size_t) j * L);
    fp_load_be<N>(Qy, g2 + (size_t) j * L + NB);
    valid &= a_first_arg_ok<N>(x, y) & a_on_curve<N>(Qx, Qy);
    uint4 *w = ws + (size_t) j * REC;
    a_ws_put<N>(w, 0, x); a_ws_put<N>(w, 1, y); a_ws_put<N>(w, 2, one); a_ws_put<N>(w, 3, one);
    a_ws_put<N>(w, 4, Qx); a_ws_put<N>(w, 5, Qy);
  }
  fp2<N> f, out;
  f.x = one;
#pragma unroll
  for (int i = 0; i < N; i++) f.y.v[i] = 0;
  a_lds_put<N>(lds_f, stride, f);
  for (int i = c_a.exp2 - 1; i >= 0; i--) {
    pbc_fair_tick<PBC_AP_FAIR_BIT>();            // (resident workgroups, pbc_hip.hip)
    a_lds_get<N>(f, lds_f, stride);
    fi_sqr<N>(f, f);
    a_lds_put<N>(lds_f, stride, f);
    for (int j = 0; j < k; j++) {
      uint4 *w = ws + (size_t) j * REC;
      jac<N> V;
      fp<N> Qx, Qy, M, XX, YY, Z3;
      a_ws_get<N>(V.X, w, 0); a_ws_get<N>(V.ZZ, w, 3);
      a_ws_get<N>(V.Y, w, 1); a_ws_get<N>(V.Z, w, 2);
      a_ws_get<N>(Qx, w, 4); a_ws_get<N>(Qy, w, 5);
      a_prod_double_head<N>(V, M, XX, YY, Z3, Qx, Qy, lds_f, stride);
      a_prod_double_tail<N>(V, M, XX, YY, Z3);
      if (i == c_a.exp1) {             // the one non-zero middle digit of r: V <- V +- P after the doubling
        fp<N> x2, y2;
        fp_load_be<N>(x2, g1 + (size_t) j * L);
        fp_load_be<N>(y2, g1 + (size_t) j * L + NB);
        if (c_a.sign1 < 0) fp_neg<N>(y2, y2);
        a_lds_get<N>(f, lds_f, stride);
        a_add_step<N>(f, V, x2, y2, Qx, Qy);
        a_lds_put<N>(lds_f, stride, f);
      }
      a_ws_put<N>(w, 0, V.X); a_ws_put<N>(w, 1, V.Y); a_ws_put<N>(w, 2, V.Z); a_ws_put<N>(w, 3, V.ZZ);
    }
  }
  a_lds_get<N>(f, lds_f, stride);
  a_final_exp<N>(out, f);
  a_store_gt<N>(gt, out, valid);
}

// ---- Type A1: the same curve y^2 = x^3 + x over a 1033-bit F_p, composite group order n --------
// a1_pairing_proj (a_param.c:1840-2015): plain double-and-add over the bits of n (tangent, double,
// [chord, add], square; the last tangent at bit 0 ends the loop), then f^(p-1) and the power by
// l = (p+1)/n (:1986-1994).  The step routines and the final exponentiation are the type a ones;
// the Lucas ladder runs over l instead of h.
// digit of the bit-by-bit Miller loops at position i: +1, -1 or 0 (wave-uniform)
PBC_DEV int a1_digit(int i) {
  return (int) ((c_a.r[i >> 5] >> (i & 31)) & 1) - (int) ((c_a.rm[i >> 5] >> (i & 31)) & 1);
}
template <int N>
PBC_DEV bool a1_miller_lane(fp2<N> &f, const uint8_t *g1, const uint8_t *g2, uint32_t *lds_q, int lds_stride) {
  const int NB = fq_bytes<N>();
  fp<N> one, Qx, Qy;
  jac<N> V;
  fp_set<N>(one, fpk<N>().one);
  fp_load_be<N>(V.X, g1);
  fp_load_be<N>(V.Y, g1 + NB);
  fp_load_be<N>(Qx, g2);
  fp_load_be<N>(Qy, g2 + NB);
  bool valid = (int) a_first_arg_ok<N>(V.X, V.Y) & (int) a_on_curve<N>(Qx, Qy);
  if constexpr (!kMemOperands<N>) {    // register-resident fields park Q in LDS between steps
#pragma unroll
    for (int k = 0; k < N; k++) {
      lds_q[k * lds_stride] = Qx.v[k];
      lds_q[(N + k) * lds_stride] = Qy.v[k];
    }
  }
  V.Z = one;
  V.ZZ = one;
  f.x = one;
  fp_sub<N>(f.y, one, one);
  for (int i = c_a.rbits - 2; i >= 0; i--) {
    if constexpr (!kMemOperands<N>) {
#pragma unroll
      for (int k = 0; k < N; k++) {
        Qx.v[k] = lds_q[k * lds_stride];
        Qy.v[k] = lds_q[(N + k) * lds_stride];
      }
    }
    a_double_step<N>(f, V, Qx, Qy, kMemOperands<N> ? lds_q : nullptr);
    const int dig = i > 0 ? a1_digit(i) : 0;
    if (dig) {                         // V <- V +- P (signed digits, hostbn.h naf_of_half)
      fp<N> x2, y2;
      fp_load_be<N>(x2, g1);
      fp_load_be<N>(y2, g1 + NB);
      if (dig < 0) fp_neg<N>(y2, y2);
      a_add_step<N>(f, V, x2, y2, Qx, Qy);
    }
  }
  return valid;
}
// element_pairing / element_prod_pairing (a1_pairings_affine, a_param.c:2100-2193: product of the
// Miller functions, one final exponentiation) for one lane
template <int N>
PBC_DEV void a1_prod_pairing_lane(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, int k, uint32_t *lds_q,
                                  int lds_stride) {
  const int L = 2 * fq_bytes<N>();
  fp2<N> F, out;
  bool valid = a1_miller_lane<N>(F, g1, g2, lds_q, lds_stride);
  for (int j = 1; j < k; j++) {
    fp2<N> f;
    valid &= a1_miller_lane<N>(f, g1 + (size_t) j * L, g2 + (size_t) j * L, lds_q, lds_stride);
    fi_mul<N>(F, F, f);
  }
  a_final_exp<N>(out, F);
  a_store_gt<N>(gt, out, valid);
}

// pairing_pp for type a1 (a1_pairing_pp_init a_param.c:1632-1726, a1_pairing_pp_apply :1728-1818):
// one table entry per doubling and per addition of the plain double-and-add loop over n, in loop
// order; [steps][3][N] words.
template <int N>
PBC_DEV bool a1_pp_init_lane(uint32_t *tab, const uint8_t *g1) {
  const int NB = fq_bytes<N>();
  fp<N> one, x2, y2, cA, cB, cC;
  jac<N> V;
  fp_set<N>(one, fpk<N>().one);
  fp_load_be<N>(x2, g1);
  fp_load_be<N>(y2, g1 + NB);
  bool valid = a_first_arg_ok<N>(x2, y2);
  V.X = x2; V.Y = y2; V.Z = one; V.ZZ = one;
  int slot = 0;
  for (int i = c_a.rbits - 2; i >= 0; i--) {
    a_pp_dbl<N>(V, cA, cB, cC);
    a_pp_store<N>(tab, slot++, cA, cB, cC);
    const int dig = i > 0 ? a1_digit(i) : 0;
    if (dig) {
      fp<N> ys = y2;
      if (dig < 0) fp_neg<N>(ys, ys);
      a_pp_add<N>(V, x2, ys, cA, cB, cC);
      a_pp_store<N>(tab, slot++, cA, cB, cC);
    }
  }
  return valid;
}
template <int N>
PBC_DEV void a1_pp_apply_lane(uint8_t *gt, const uint32_t *tab, bool p_valid, const uint8_t *g2, uint32_t *lds_f = nullptr) {
  const int NB = fq_bytes<N>();
  fp<N> Qx, Qy;
  fp2<N> f, out;
  fp<N> g_private[2];                  // f^2 of a step: in the lane's LDS slots on the device (6 of a step's 26 element moves)
  fp2v<N> g{(kMemOperands<N> && lds_f) ? *reinterpret_cast<fp<N> *>(lds_f) : g_private[0],
            (kMemOperands<N> && lds_f) ? *reinterpret_cast<fp<N> *>(lds_f + 36) : g_private[1]};
  fp_load_be<N>(Qx, g2);
  fp_load_be<N>(Qy, g2 + NB);
  bool valid = (int) p_valid & (int) a_on_curve<N>(Qx, Qy);
  fp_set<N>(f.x, fpk<N>().one);
  fp_sub<N>(f.y, f.x, f.x);
  int slot = 0;
  for (int i = c_a.rbits - 2; i >= 0; i--) {
    if constexpr (kMemOperands<N>) {
      fi_sqr_x<N>(g, f);
      a_pp_line_x<N>(f, g, tab, slot++, Qx, Qy);
      if (i > 0 && a1_digit(i)) {
        a_pp_line_x<N>(g, f, tab, slot++, Qx, Qy);
        f.x = g.x;
        f.y = g.y;
      }
      continue;
    }
    fi_sqr<N>(f, f);
    a_pp_line<N>(f, tab, slot++, Qx, Qy);
    if (i > 0 && a1_digit(i)) a_pp_line<N>(f, tab, slot++, Qx, Qy);
  }
  a_final_exp<N>(out, f);
  a_store_gt<N>(gt, out, valid);
}

}  // namespace pbc
