// pairing_f.cuh -- Type F (Barreto-Naehrig curve y^2 = x^3 + b over F_q, embedding degree 12)
// reduced Tate pairing, one pairing per lane.
//
// Computes the same GT value as the reference's f_pairing (ecc/f_param.c:289-311):
// untwisting by negalphainv, cc_miller_no_denom (:97-248) with the fused sparse line
// multiplication f_miller_evalfn (:109-149), and f_tateexp (:250-283: Frobenius easy part,
// then the (q^4 - q^2 + 1)/r power).  Towers as in the reference so that the output
// serialises identically:  F_q^2 = F_q[sqrt(beta)] (fieldquadratic.c fq_*),
// F_q^12 = F_q^2[x]/(x^6 + alpha) with basis 1, x, ..., x^5 (poly.c polymod, n = 6).
//
// GPU re-design:
//   * Jacobian Miller loop on E(F_q) (the reference's affine loop: one F_q inversion per
//     step, 215 per pairing); projective line coefficients differ by F_q^*
//     factors which (q^12-1)/r removes;
//   * the sparse line (a Qx x^4 + b Qy x^3 + c) is pre-multiplied once per step
//     (a Qx, b Qy and their products with -alpha), then each output coefficient is two
//     F_q^2 products and one F_q scaling;
//   * an F_q^12 element is 60 words: it lives in per-lane private memory and the tower
//     routines are rolled loops over coefficients (compact code, instruction-cache
//     resident); all arithmetic happens on registers in F_q / F_q^2 granules;
//   * the hard part of the final exponentiation uses a fixed 4-bit window.
#pragma once
#include "fp.cuh"

#ifndef PBC_F_FAIR_BIT
#define PBC_F_FAIR_BIT 21                 // slices of 2^21 cycles (0.9 ms): 28.4 ms per 2^18 launch; 2^16: 30.3, 2^19: 28.8, 2^23: 28.8, none: 31.2
#endif
// Where a filling wide accumulator is relieved by wide_squeeze (fp.cuh: a few shifts) instead of a Montgomery reduction:
// bit 1: f12_sqr_lds, 2: f12_mul_lds, 4: the private-memory coefficient bodies (there: only where the instantiation does
// not keep the beta y_i array -- PBC_F_BY bit 4 -- or the field has more than 6 limbs).
// Same-box A/B on f.param / the 254-bit field (ms per 2^18 / 2^16 launch; profiles/r03_notes.md), with the beta y_i
// arrays still kept: none 29.29 / 53.9, 1: 28.63, 1 + 4: 30.53 / 53.3, 1 + 2 + 4: 32.03 / 52.0, 4: 31.49 / 50.6 -- on the
// 5-word fields the compiler answered the change in f12_mul_lds and the private bodies with 3.5 x / 1.6 x the spill
// traffic, which cost more than the 4-5 % fewer multiply-adds bring.  With beta y_i recomputed (PBC_F_BY 7) the private
// bodies take it: 26.9 -> 26.1 ms (f12_mul: 83 -> 25 scratch instructions); f12_mul_lds still answers with spills.
#ifndef PBC_F_SQZ
#define PBC_F_SQZ (1 | 4)
#endif
// Where beta y_i of the i-basis kernels (a negation and a carry pass: 24 instructions) is recomputed when a pair needs it
// instead of kept for all six coefficients -- bit 1: f12_sqr_lds, 2: f12_mul_lds, 4: the private-memory bodies.  The
// kept array ends up in private memory (f12_sqr_lds read it back by address: 63 of the squaring's 83 scratch
// instructions).  Same-box A/B on f.param 2^18, ms per launch: 0: 28.5, 1: 27.55, 3: 27.35, 5: 27.07, 7: 26.9.
#ifndef PBC_F_BY
#define PBC_F_BY 7
#endif
#ifndef PBC_F_PREFETCH
#define PBC_F_PREFETCH 1               // one-area layout: the buffered result coefficients are read back before the last ones are computed
#endif
#ifndef PBC_F_NO_CONJ
#define PBC_F_NO_CONJ 0                // 1: the q^6-power Frobenius through the generic qpower routine (A/B of round 5's f12_conj)
#endif
#ifndef PBC_F_LINE_SEL
#define PBC_F_LINE_SEL 1               // f_line_mul_lds: the factors of an output switched by two branches on the loop counter (0: 36 selects in every iteration; same-box A/B 22.51 / 22.62 -> 22.48 / 22.44 ms, profiles/r06_notes.md)
#endif
#ifdef PBC_HOSTSIM
#define PBC_KEEP_BRANCH() ((void) 0)
#else
#define PBC_KEEP_BRANCH() asm volatile("" ::: "memory")     // (keeps the compiler from turning a wave-uniform branch back into per-lane selects)
#endif
#ifndef PBC_F_LINE_LIMB
#define PBC_F_LINE_LIMB 1              // f_line_mul_lds: the line's pre-multiplication by Q and -alpha in limb form (0: word-form calls)
#endif
namespace pbc {

constexpr int ND = 5;                  // the 158-bit BN field of f.param: 5 x 32-bit words, 6 x 29-bit limbs
constexpr int NF_MAX = 8;              // widest BN field built in: 256 bits (pbc_param_init_f_gen(256))
struct FConst {                        // f_pairing_data_s (ecc/f_param.c:35-45)
  uint32_t B[NF_MAX];                  // curve b (Montgomery form)
  uint32_t beta[NF_MAX];               // nqr of F_q (f_param.c:345-348)
  uint32_t negalpha[2][NF_MAX];        // X^6 (f_param.c:355-361)
  uint32_t negalphainv[2][NF_MAX];
  uint32_t xpowq2[2][NF_MAX], xpowq6[2][NF_MAX], xpowq8[2][NF_MAX];   // X^(q^k) = (this) X (f_param.c:431-444)
  uint32_t tb[2][NF_MAX];              // twist: y^2 = x^3 - alpha b (f_param.c:372-381)
  uint32_t r[9], rm[9];                // Miller loop digits: NAF of r >> 1, +1 digits in r[], -1 digits in rm[] (hostbn.h); a 256-bit r can put its leading digit at position 256
  uint32_t tateexp[32];                // (q^4 - q^2 + 1)/r (f_param.c:414-420)
  int rbits, tebits;
  // BN structure (f_param.c:70-95 tryplusx/tryminusx): q = 36x^4+36x^3+24x^2+6x+1,
  // r = 36x^4+36x^3+18x^2+6x+1.  When the host recognises it, the hard part uses the
  // x-chain instead of a 472-bit power.
  // limb forms (29-bit) of the constants the lazy F_q^12 products multiply by: read as scalar operands, no conversions
  uint32_t beta29[9], one29[9], na29[2][9], bna29[9];   // beta, R mod q, negalpha (x, y), beta * negalpha.y
  uint32_t gamma[2][NF_MAX];           // X^q = gamma X, gamma = negalpha^((q-1)/6)
  uint32_t bn_x[2];                    // |x|
  int bn_ok, bn_xneg, bn_xbits;
  // "i-basis" copy of the constants (q = 3 mod 4, pairing kernels only; see init_stage3): F_q^2 is represented as
  // F_q[i], i^2 = -1, through s -> c i with c^2 = -beta, so that multiplying by beta is a negation
  int bm1;                             // 1: beta = -1 in this block
  uint32_t cmap[NF_MAX], cinv[NF_MAX]; // c, 1/c (Montgomery form): (x, y) -> (x, c y) on the way in, (x, y / c) on the way out
  uint32_t kneg29[9];                  // 4 q in borrowed limb form (hostbn.h ksub_build, D = 1): K - y is -y with non-negative limbs
  // cyclotomic squarings of the hard part (f12_cyc_sqr_lds): 3 xi.x, 3 xi.y, 3 beta xi.y, 6 xi.x, 6 xi.y, 6 beta xi.y for
  // xi = negalpha as limbs, and 8 q in borrowed limbs with D = 2 (K - 2 y for normalised y); cyc_ok: both K constants fit q
  uint32_t cyc29[6][9], kneg8_29[9];
  int cyc_ok;
  // point arithmetic on E(F_q) in limb form (six-limb fields; f_dbl_core_l / f_add_core_l): c q in borrowed limbs for
  // (c, D) = (2, 1), (4, 2), (16, 2), (32, 2) -- the constants of pairing_d.cuh's limb-form steps; pl_ok: they fit this q
  uint32_t pk29[4][6];
  int pl_ok;
  // A sparse xi (init_stage4; pairing kernels of the 5-word fields in the i-basis): F_q^12 is taken in the basis 1, X', ...,
  // X'^5 with X' = X / c, X'^6 = xi' = xs_u + xs_v i for SMALL integers xs_u, xs_v (c^6 = xi / xi'; xi' lies in xi's class
  // of F_q^2* modulo 6th powers).  Every constant of this block is then the X'-basis one (negalpha = xi', gamma = xi'^((q-1)/6),
  // ...), so the tower routines run unchanged; products by xi' are a few shifts and adds where the routines know it
  // (TypeF<.., XS>).  The Miller loop scales Q by c^-2 / c^-3 on the way in and coefficient i of the result by c^-i on
  // the way out.  xs_ok = 0: the parameter file's xi, c = 1.
  uint32_t xc_inv[2][NF_MAX];          // 1 / c
  int xs_ok, xs_u, xs_v;               // xs_u in {1, 2, 4}, xs_v in {1, 2}
  int xs_su, xs_sv;                    // their binary logarithms
};
static_assert(sizeof(FConst) <= KOFF_XS - KOFF_TYPE, "constant block layout");
#define c_f (pbc::kconst<pbc::FConst, pbc::KOFF_TYPE>())
struct FRaw {
  uint32_t b[NF_MAX], beta[NF_MAX], alpha0[NF_MAX], alpha1[NF_MAX];
  uint32_t e6[NF_MAX + 1];
  int e6bits;
  uint32_t e4[NF_MAX + 1];             // (q + 1) / 4 when q = 3 mod 4 and the i-basis is wanted (e4bits > 0), with 4 q for kneg29
  int e4bits;
  uint32_t kneg29[9], kneg8_29[9];     // 4 q (D = 1) and 8 q (D = 2) in borrowed limbs; k_ok: both fit this q (hostbn.h ksub_build)
  int k_ok;
  uint32_t pk29[4][6];                 // 2 q (D = 1), 4 q, 16 q, 32 q (D = 2) for the limb-form point arithmetic; pl_ok: all fit
  int pl_ok;
  // init_stage4 (a sparse xi): (q^2 - 1) / 6, m, m u, S v t mod (q^2 - 1) and S (host_params.h init_type_f)
  uint32_t xs_ecls[11], xs_m[11], xs_eS[11], xs_em[11];
  int xs_eclsbits, xs_mbits, xs_eSbits, xs_embits, xs_S, xs_try;
};

// LDS staging area of the F_q^12 products: the limb forms (x, y of six coefficients) of one operand per lane,
// limb-major for 128-lane workgroups (36 KB for the 5-word field: four workgroups per CU)
// (5-word fields: two such areas, so that the Miller accumulator can be updated from one into the other)
#ifndef PBC_F_AREAS
#define PBC_F_AREAS 1                  // LDS areas of the 5-word fields: 1 (36 KB per workgroup, two waves per SIMD) or 2 (72 KB, one)
#endif
template <int ND> constexpr int kF12Bufs = Limbs29<ND>::L <= 6 ? PBC_F_AREAS : 1;
template <int ND> __shared__ uint32_t g_lds_f12[kF12Bufs<ND> * 12 * Limbs29<ND>::L * 128];

// Everything below is per field width: ND 32-bit words per F_q element (5 for f.param, 8 for 256-bit BN fields).
// BM1: the instantiation of the pairing kernels for objects with i-basis constants (FConst::bm1, init_stage3)
// XS: the instantiation for blocks with a sparse xi (FConst::xs_ok, init_stage4): products by xi' are shifts and adds
template <int ND, bool BM1 = false, bool XS = false>
struct TypeF {
static_assert(!XS || BM1, "the sparse xi lives in the i-basis");
typedef fp<ND> fq;
typedef typename vecN<ND>::type v5;
static PBC_DEV fq dk(const uint32_t *w) { fq r; fp_set<ND>(r, w); return r; }
static PBC_DEV int fb() { return (int) fpk<ND>().fbytes; }

struct g2 { fq x, y; };                // x + y sqrt(beta)
struct f12 { g2 c[6]; };               // sum c_i X^i, X^6 = negalpha
struct djac { fq X, Y, Z, ZZ; };

static PBC_DEV fl<ND> fl29(const uint32_t *l) {                 // a constant kept as limbs in the constant block
  fl<ND> r;
#pragma unroll
  for (int i = 0; i < Limbs29<ND>::L; i++) r.l[i] = l[i];
  return r;
}
static PBC_DEV void fair_tick() { pbc_fair_tick<PBC_F_FAIR_BIT>(); }
// beta * y on limb forms (y: normalised limbs, value < 2.001 q).  General beta: a Montgomery product with the constant.
// i-basis (BM1): K - y with K = 4 q in borrowed limbs, then a parallel carry pass -- 24 instructions
// instead of 72 multiply-adds; the result has limbs <= 2^29 + 6 and a value below 4 q, which every sum it enters holds.
static PBC_DEV void mul_beta(fl<ND> &r, const fl<ND> &y) {
  constexpr int L = Limbs29<ND>::L;
  if constexpr (BM1) {
    uint32_t t[L], c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) t[i] = c_f.kneg29[i] - y.l[i];
#pragma unroll
    for (int i = 0; i < L; i++) {
      r.l[i] = (i < L - 1 ? (t[i] & Limbs29<ND>::MASK) : t[i]) + c;
      c = t[i] >> 29;
    }
  } else {
    const fl<ND> xx[1] = {y}, yy[1] = {fl29(c_f.beta29)};
    sop_limbs<ND, 1>(r, xx, yy);
  }
}
// xi' (x + y i) for the sparse xi' = u + v i of init_stage4 (u in {1, 2, 4}, v in {1, 2}; i-basis): (u x - v y) + (v x + u y) i
// by shifts, the borrowed constant K = 8 q (which dominates 2 y for normalised y) and a parallel carry pass: 50
// instructions instead of the 144 multiply-adds and two reductions of a product by a general xi.  In: normalised limbs,
// values below 2.001 q; out: limbs <= 2^29 + 8, values below 16 q / 12 q, which every sum they enter holds.
static PBC_DEV void mul_xi(fl<ND> &rx, fl<ND> &ry, const fl<ND> &x, const fl<ND> &y) {
  constexpr int L = Limbs29<ND>::L;
  const int su = c_f.xs_su, sv = c_f.xs_sv;
  uint32_t t[L], u[L], c = 0, d = 0;
#pragma unroll
  for (int i = 0; i < L; i++) {
    t[i] = (x.l[i] << su) + c_f.kneg8_29[i] - (y.l[i] << sv);
    u[i] = (x.l[i] << sv) + (y.l[i] << su);
  }
#pragma unroll
  for (int i = 0; i < L; i++) {
    rx.l[i] = (i < L - 1 ? (t[i] & Limbs29<ND>::MASK) : t[i]) + c;
    c = t[i] >> 29;
    ry.l[i] = (i < L - 1 ? (u[i] & Limbs29<ND>::MASK) : u[i]) + d;
    d = u[i] >> 29;
  }
}
static PBC_DEV g2 fk2(const uint32_t (*w)[NF_MAX]) { g2 r; fp_set<ND>(r.x, w[0]); fp_set<ND>(r.y, w[1]); return r; }

// ---- F_q^2 = F_q[sqrt(beta)] -------------------------------------------------------------
static PBC_DEV void g2_add(g2 &r, const g2 &a, const g2 &b) { fp_add<ND>(r.x, a.x, b.x); fp_add<ND>(r.y, a.y, b.y); }
static PBC_DEV void g2_sub(g2 &r, const g2 &a, const g2 &b) { fp_sub<ND>(r.x, a.x, b.x); fp_sub<ND>(r.y, a.y, b.y); }
static PBC_DEV void g2_dbl(g2 &r, const g2 &a) { fp_dbl<ND>(r.x, a.x); fp_dbl<ND>(r.y, a.y); }
static PBC_DEV void g2_neg(g2 &r, const g2 &a) { fp_neg<ND>(r.x, a.x); fp_neg<ND>(r.y, a.y); }
static PBC_DEV void g2_mul_fq(g2 &r, const g2 &a, const fq &s) { fp_mul<ND>(r.x, a.x, s); fp_mul<ND>(r.y, a.y, s); }
static PBC_DEV bool g2_eq(const g2 &a, const g2 &b) { return fp_eq<ND>(a.x, b.x) & fp_eq<ND>(a.y, b.y); }
static PBC_DEV void g2_zero(g2 &r) {
#pragma unroll
  for (int k = 0; k < ND; k++) { r.x.v[k] = 0; r.y.v[k] = 0; }
}
// fq_mul (fieldquadratic.c:197-233), lazily reduced:  t = beta a.y;
//   re = a.x b.x + t b.y,  im = a.x b.y + a.y b.x     (5 limb products, 3 reductions)
static PBC_DEV void g2_mul_inl(g2 &r, const g2 &a, const g2 &b) {
  fl<ND> ax, ay, bx, by, t, c;
  to_limbs<ND>(ax, a.x); to_limbs<ND>(ay, a.y);
  to_limbs<ND>(bx, b.x); to_limbs<ND>(by, b.y);
  mul_beta(t, ay);
  { const fl<ND> x[2] = {ax, t}, y[2] = {bx, by}; sop_limbs<ND, 2>(c, x, y); from_limbs<ND>(r.x, c); }
  { const fl<ND> x[2] = {ax, ay}, y[2] = {by, bx}; sop_limbs<ND, 2>(c, x, y); from_limbs<ND>(r.y, c); }
}
// fq_square (fieldquadratic.c:249-269): re = a.x^2 + beta a.y^2, im = 2 a.x a.y
static PBC_DEV void g2_sqr_inl(g2 &r, const g2 &a) {
  fl<ND> ax, ay, ax2, t, c;
  to_limbs<ND>(ax, a.x); to_limbs<ND>(ay, a.y);
  limbs_dbl<ND>(ax2, ax);
  mul_beta(t, ay);
  { const fl<ND> x[2] = {ax, t}, y[2] = {ax, ay}; sop_limbs<ND, 2>(c, x, y); from_limbs<ND>(r.x, c); }
  { const fl<ND> x[1] = {ax2}, y[1] = {ay}; sop_limbs<ND, 1, 1>(c, x, y); from_limbs<ND>(r.y, c); }
}
struct g2ret { v5 x, y; };
static __device__ __noinline__ g2ret g2_mul_call(v5 ax, v5 ay, v5 bx, v5 by) {
  g2 a, b, r;
  from_vec<ND>(a.x, ax); from_vec<ND>(a.y, ay); from_vec<ND>(b.x, bx); from_vec<ND>(b.y, by);
  g2_mul_inl(r, a, b);
  return g2ret{to_vec<ND>(r.x), to_vec<ND>(r.y)};
}
static __device__ __noinline__ g2ret g2_sqr_call(v5 ax, v5 ay) {
  g2 a, r;
  from_vec<ND>(a.x, ax); from_vec<ND>(a.y, ay);
  g2_sqr_inl(r, a);
  return g2ret{to_vec<ND>(r.x), to_vec<ND>(r.y)};
}
static PBC_DEV void g2_mul(g2 &r, const g2 &a, const g2 &b) {
  g2ret t = g2_mul_call(to_vec<ND>(a.x), to_vec<ND>(a.y), to_vec<ND>(b.x), to_vec<ND>(b.y));
  from_vec<ND>(r.x, t.x); from_vec<ND>(r.y, t.y);
}
static PBC_DEV void g2_sqr(g2 &r, const g2 &a) {
  g2ret t = g2_sqr_call(to_vec<ND>(a.x), to_vec<ND>(a.y));
  from_vec<ND>(r.x, t.x); from_vec<ND>(r.y, t.y);
}
// xi' a for the sparse xi' = u + v i (XS instantiation; word form): (u x - v y) + (v x + u y) i by doublings
static PBC_DEV void g2_mul_xi(g2 &r, const g2 &a) {
  fq ux = a.x, uy = a.y, vx = a.x, vy = a.y;
  for (int k = 0; k < c_f.xs_su; k++) { fp_dbl<ND>(ux, ux); fp_dbl<ND>(uy, uy); }
  for (int k = 0; k < c_f.xs_sv; k++) { fp_dbl<ND>(vx, vx); fp_dbl<ND>(vy, vy); }
  fp_sub<ND>(r.x, ux, vy);
  fp_add<ND>(r.y, vx, uy);
}
// fq_invert (fieldquadratic.c:290-309)
static PBC_DEV void g2_inv(g2 &r, const g2 &a) {
  fq e0, e1;
  fp_sqr<ND>(e0, a.x);
  fp_sqr<ND>(e1, a.y);
  fp_mul<ND>(e1, e1, dk(c_f.beta));
  fp_sub<ND>(e0, e0, e1);
  fp_inv<ND>(e0, e0);
  fp_mul<ND>(r.x, a.x, e0);
  fp_neg<ND>(e0, e0);
  fp_mul<ND>(r.y, a.y, e0);
}
static PBC_DEV void g2_load_be(g2 &r, const uint8_t *s) { fp_load_be<ND>(r.x, s); fp_load_be<ND>(r.y, s + fb()); }
static PBC_DEV void g2_store_be(uint8_t *d, const g2 &a) { fp_store_be<ND>(d, a.x); fp_store_be<ND>(d + fb(), a.y); }

// ---- F_q^12 = F_q^2[X]/(X^6 + alpha): rolled loops over private-memory coefficients -------
static __device__ __noinline__ void f12_one(f12 *r) {
  fq one;
  fp_set<ND>(one, fpk<ND>().one);
#pragma nounroll
  for (int i = 0; i < 6; i++) g2_zero(r->c[i]);
  r->c[0].x = one;
}
// ---- products in F_q^12: one operand in registers, the other staged in LDS ---------------------------------
// An F_q^12 element is 60 words and lives in the lane's private memory between operations.  The product routines do
// not work on that memory: the first operand's limb forms (x, y and beta*y of the six coefficients) are loaded into
// registers once, with compile-time indices; the second operand's limb forms are staged in LDS (limb-major,
// [coefficient][x|y][limb][lane]: conflict-free), where the partner coefficient of a product -- whose index depends on the
// rolled loop over the output coefficients -- is read at LDS speed.  Per product the private memory sees the operands
// once and the result once (about 200 dword accesses instead of more than a thousand).
static constexpr int FL = Limbs29<ND>::L;
static constexpr int F_LANES = 128;
static PBC_DEV fl<ND> ldsf_get(int c, int part, int buf = 0) {
  fl<ND> r;
#pragma unroll
  for (int i = 0; i < FL; i++) r.l[i] = g_lds_f12<ND>[(((buf * 6 + c) * 2 + part) * FL + i) * F_LANES + threadIdx.x];
  return r;
}
static PBC_DEV void ldsf_put(int c, int part, const fl<ND> &a, int buf = 0) {
#pragma unroll
  for (int i = 0; i < FL; i++) g_lds_f12<ND>[(((buf * 6 + c) * 2 + part) * FL + i) * F_LANES + threadIdx.x] = a.l[i];
}
template <int BIT> static constexpr bool kByRe = BM1 && (PBC_F_BY & BIT) != 0;
template <bool WITH_BY = true>
struct f12r_t { fl<ND> x[6], y[6], by[WITH_BY ? 6 : 1]; };     // register-resident operand: every access uses a compile-time index
typedef f12r_t<!kByRe<4>> f12r;                 // (the private-memory bodies)
// a -> registers (and, when `stage`, its x / y limb forms to LDS as well: the squaring's second operand is the first)
template <bool WITH_BY>
static PBC_DEV void f12_load_regs(f12r_t<WITH_BY> &A, const f12 *a, bool stage) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    to_limbs<ND>(A.x[i], a->c[i].x);
    to_limbs<ND>(A.y[i], a->c[i].y);
    if constexpr (WITH_BY) mul_beta(A.by[i], A.y[i]);
    if (stage) { ldsf_put(i, 0, A.x[i]); ldsf_put(i, 1, A.y[i]); }
  }
}
static PBC_DEV void f12_stage(const f12 *b) {
#pragma nounroll
  for (int i = 0; i < 6; i++) {
    fl<ND> x, y;
    to_limbs<ND>(x, b->c[i].x);
    to_limbs<ND>(y, b->c[i].y);
    ldsf_put(i, 0, x);
    ldsf_put(i, 1, y);
  }
}
template <bool WITH_BY>
static PBC_DEV fl<ND> f12r_by(const f12r_t<WITH_BY> &A, int i) {
  if constexpr (WITH_BY) return A.by[i];
  else { fl<ND> t; mul_beta(t, A.y[i]); return t; }
}
// capacity of a wide accumulator in product units: (units + 1) L 2^58 < 2^64
static constexpr int kCap = 63 / Limbs29<ND>::L - 1;
static_assert(2 * 6 + 2 + 1 <= kWideMaxUnits, "a coefficient of a product or square: at most six pairs + the fold");
static constexpr bool kSqueezePriv = (PBC_F_SQZ & 4) != 0 && ((BM1 && (PBC_F_BY & 4) != 0) || Limbs29<ND>::L > 6);
static_assert(kCap >= 6, "field too wide for the F_q^12 accumulators");
static PBC_DEV void wide_flush(g2 &acc, wide<ND> &Wx, wide<ND> &Wy) {
  fl<ND> t;
  fq u;
  wide_reduce<ND>(t, Wx); from_limbs<ND>(u, t); fp_add<ND>(acc.x, acc.x, u);
  wide_reduce<ND>(t, Wy); from_limbs<ND>(u, t); fp_add<ND>(acc.y, acc.y, u);
  wide_zero<ND>(Wx);
  wide_zero<ND>(Wy);
}
// polymod_mul (poly.c:1005-1047): schoolbook over the coefficients.  Coefficient k of the plain product:
//     re = sum_{i+j=k} a_i.x b_j.x + (beta a_i.y) b_j.y,   im = sum_{i+j=k} a_i.x b_j.y + a_i.y b_j.x
// accumulated UNREDUCED in wide column accumulators: one Montgomery reduction per component of a coefficient (the
// columns are relieved by wide_squeeze when they fill).  A in registers, b_j from LDS.
static PBC_DEV g2 f12_mul_coeff(const f12r &A, int k) {
  g2 acc;
  g2_zero(acc);
  wide<ND> Wx, Wy;
  wide_zero<ND>(Wx);
  wide_zero<ND>(Wy);
  int units = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int j = k - i;
    if (j < 0 || j > 5) continue;                // wave-uniform
    if (units + 2 > kCap) { if constexpr (kSqueezePriv) { wide_squeeze<ND>(Wx); wide_squeeze<ND>(Wy); units = 1; } else { wide_flush(acc, Wx, Wy); units = 0; } }
    const fl<ND> bx = ldsf_get(j, 0), by = ldsf_get(j, 1);
    wide_mac<ND>(Wx, A.x[i], bx);
    wide_mac<ND>(Wx, f12r_by(A, i), by);
    wide_mac<ND>(Wy, A.x[i], by);
    wide_mac<ND>(Wy, A.y[i], bx);
    units += 2;
  }
  wide_flush(acc, Wx, Wy);
  return acc;
}
// polymod_square (poly.c:1091-1143): cross terms once with a doubled operand, squares once.
// A cross pair is 2 doubled products per accumulator (4 capacity units), a square pair 2 units.
static PBC_DEV g2 f12_sqr_coeff(const f12r &A, int k) {
  g2 acc;
  g2_zero(acc);
  wide<ND> Wx, Wy;
  wide_zero<ND>(Wx);
  wide_zero<ND>(Wy);
  int units = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int j = k - i;
    if (j < i || j > 5) continue;                // pairs i <= j, wave-uniform
    if (units + 4 > kCap) { if constexpr (kSqueezePriv) { wide_squeeze<ND>(Wx); wide_squeeze<ND>(Wy); units = 1; } else { wide_flush(acc, Wx, Wy); units = 0; } }
    if (i == j) {
      // a_i^2: re = x^2 + (beta y) y, im = 2 x y
      fl<ND> ax2;
      limbs_dbl<ND>(ax2, A.x[i]);
      wide_mac<ND>(Wx, A.x[i], A.x[i]);
      wide_mac<ND>(Wx, f12r_by(A, i), A.y[i]);
      wide_mac<ND>(Wy, ax2, A.y[i]);
      units += 2;
    } else {
      // 2 a_i a_j
      fl<ND> bx2, by2;
      limbs_dbl<ND>(bx2, ldsf_get(j, 0));
      limbs_dbl<ND>(by2, ldsf_get(j, 1));
      wide_mac<ND>(Wx, A.x[i], bx2);
      wide_mac<ND>(Wx, f12r_by(A, i), by2);
      wide_mac<ND>(Wy, A.x[i], by2);
      wide_mac<ND>(Wy, A.y[i], bx2);
      units += 4;
    }
  }
  wide_flush(acc, Wx, Wy);
  return acc;
}
// r_k = d_k + negalpha d_{k+6}  (X^(6+k) = negalpha X^k), k = 0..5; d_11 does not exist
// (the two coefficients of a fold pair share one instance of the coefficient body: instruction-cache footprint)
static __device__ __noinline__ void f12_mul(f12 *r, const f12 *a, const f12 *b) {
  fair_tick();
  f12r A;
  f12_load_regs(A, a, false);
  f12_stage(b);
  const g2 na = fk2(c_f.negalpha);
#pragma nounroll
  for (int k = 0; k < 6; k++) {
    g2 t;
    g2_zero(t);
#pragma nounroll
    for (int h = 1; h >= 0; h--) {
      if (h && k == 5) continue;
      g2 u = f12_mul_coeff(A, k + 6 * h);
      if (h) { if constexpr (XS) g2_mul_xi(t, u); else g2_mul(t, u, na); } else g2_add(t, t, u);
    }
    r->c[k] = t;                                 // r may be a or b: both are in registers / LDS by now
  }
}
static __device__ __noinline__ void f12_sqr(f12 *r, const f12 *a) {
  fair_tick();
  f12r A;
  f12_load_regs(A, a, true);
  const g2 na = fk2(c_f.negalpha);
#pragma nounroll
  for (int k = 0; k < 6; k++) {
    g2 t;
    g2_zero(t);
#pragma nounroll
    for (int h = 1; h >= 0; h--) {
      if (h && k == 5) continue;
      g2 u = f12_sqr_coeff(A, k + 6 * h);
      if (h) { if constexpr (XS) g2_mul_xi(t, u); else g2_mul(t, u, na); } else g2_add(t, t, u);
    }
    r->c[k] = t;
  }
}
// coefficient-wise even-power Frobenius: out^(q^k), X^(q^k) = e X (qpower, f_param.c:257-268)
static __device__ __noinline__ void f12_qpower(f12 *r, const f12 *a, const uint32_t (*ew)[NF_MAX]) {
  const g2 e = fk2(ew);
  g2 epow = e;
  r->c[0] = a->c[0];
#pragma nounroll
  for (int i = 1; i < 6; i++) {
    g2 t;
    g2_mul(t, a->c[i], epow);
    r->c[i] = t;
    g2_mul(epow, epow, e);
  }
}
// The q^6-power Frobenius (qpower with xpowq6, f_param.c:257-268, 441): X^(q^6) = xi^((q^6 - 1) / 6) X, and
// xi^((q^6 - 1) / 6) = (xi^((q^2 - 1) / 2))^((q^4 + q^2 + 1) / 3) = -1 because xi is a non-square of F_q^2 (X^6 - xi is
// irreducible) and (q^4 + q^2 + 1) / 3 is odd -- the same in the basis X' = X / c of a sparse xi (c lies in F_q^2).  So the
// map is "negate the odd coefficients": six F_q negations instead of the ten F_q^2 products of the generic routine
// (2.1 k multiply-adds; seven calls per pairing, one per window of a GT power).  r may be a.
static PBC_DEV void f12_conj(f12 *r, const f12 *a) {
#if PBC_F_NO_CONJ
  f12_qpower(r, a, c_f.xpowq6);        // (A/B switch: the generic routine, rounds 1-4)
  return;
#endif
#pragma nounroll
  for (int i = 0; i < 6; i++) {
    g2 t = a->c[i];
    if (i & 1) g2_neg(t, t);
    r->c[i] = t;
  }
}
// polymod_invert (poly.c:521-536): unique inverse; sigma = q^2-power Frobenius,
// a^-1 = prod_{i=1..5} sigma^i(a) / N,  N = a * prod in F_q^2
static __device__ __noinline__ void f12_inv(f12 *r, const f12 *a) {
  f12 s, t, n;
  f12_qpower(&s, a, c_f.xpowq2);
  t = s;
#pragma nounroll
  for (int i = 2; i <= 5; i++) {
    f12_qpower(&s, &s, c_f.xpowq2);
    f12_mul(&t, &t, &s);
  }
  f12_mul(&n, a, &t);
  g2 ni;
  g2_inv(ni, n.c[0]);
#pragma nounroll
  for (int i = 0; i < 6; i++) {
    g2 u;
    g2_mul(u, t.c[i], ni);
    r->c[i] = u;
  }
}

// v <- v * (a Qx X^4 + b Qy X^3 + c)   (f_miller_evalfn, f_param.c:109-149)
// out_i = c v_i + [aQx] v_{i-4} + [bQy] v_{i-3}, indices mod 6 with a factor negalpha on wrap.
// v is staged in LDS; each output coefficient is ONE lazily reduced sum per component (five limb products, one
// Montgomery reduction):
//   re = c v_i.x + fa.x v_j.x + (beta fa.y) v_j.y + fb.x v_k.x + (beta fb.y) v_k.y
//   im = c v_i.y + fa.x v_j.y + fa.y v_j.x + fb.x v_k.y + fb.y v_k.x
// (a, b, c travel as vectors: by-value fq structs beyond clang's 16-register aggregate budget
// are passed indirectly, and that path miscompiled here -- see profiles/r01_notes.md)
struct g2l { fl<ND> x, y, by; };
static PBC_DEV void g2l_make(g2l &r, const g2 &a) {
  to_limbs<ND>(r.x, a.x);
  to_limbs<ND>(r.y, a.y);
  mul_beta(r.by, r.y);
}
// (x + y s) s for an F_q scalar, and a product by the constant -alpha, on limb forms (values below 2q in, below 2q out)
template <int DBL = 0>                 // DBL 1: the scalar's limbs reach 2^30
static PBC_DEV void g2l_scale(g2l &r, const fl<ND> &x, const fl<ND> &y, const fl<ND> &s) {
  { const fl<ND> u[1] = {x}, v[1] = {s}; sop_limbs<ND, 1, DBL>(r.x, u, v); }
  { const fl<ND> u[1] = {y}, v[1] = {s}; sop_limbs<ND, 1, DBL>(r.y, u, v); }
  mul_beta(r.by, r.y);
}
static PBC_DEV void g2l_mul_na(g2l &r, const g2l &a) {
  if constexpr (XS) {                  // sparse xi: shifts and adds
    mul_xi(r.x, r.y, a.x, a.y);
    // beta (xi' a).y = -(v x + u y) = v (-x) + u (-y) from the negations of NORMALISED values (mul_beta's K = 4 q does not
    // dominate the 12 q that (xi' a).y can reach): limbs <= 2^29 + 8, value below 24 q
    fl<ND> nx;
    mul_beta(nx, a.x);
    uint32_t t[FL], c = 0;
#pragma unroll
    for (int i = 0; i < FL; i++) t[i] = (nx.l[i] << c_f.xs_sv) + (a.by.l[i] << c_f.xs_su);
#pragma unroll
    for (int i = 0; i < FL; i++) {
      r.by.l[i] = (i < FL - 1 ? (t[i] & Limbs29<ND>::MASK) : t[i]) + c;
      c = t[i] >> 29;
    }
    return;
  }
  const fl<ND> nax = fl29(c_f.na29[0]), nay = fl29(c_f.na29[1]);
  { const fl<ND> u[2] = {a.x, a.by}, v[2] = {nax, nay}; sop_limbs<ND, 2>(r.x, u, v); }
  { const fl<ND> u[2] = {a.x, a.y}, v[2] = {nay, nax}; sop_limbs<ND, 2>(r.y, u, v); }
  mul_beta(r.by, r.y);
}
static PBC_DEV void g2l_sel(g2l &r, const g2l &a, const g2l &b, bool take_b) {
#pragma unroll
  for (int i = 0; i < FL; i++) {
    r.x.l[i] = take_b ? b.x.l[i] : a.x.l[i];
    r.y.l[i] = take_b ? b.y.l[i] : a.y.l[i];
    r.by.l[i] = take_b ? b.by.l[i] : a.by.l[i];
  }
}
static_assert(kCap >= 5, "five products per accumulator in f_line_mul");
static __device__ __noinline__ void f_line_mul(f12 *v, v5 va, v5 vb, v5 vc, const g2 *Qx, const g2 *Qy) {
  fq a, b, c;
  from_vec<ND>(a, va);
  from_vec<ND>(b, vb);
  from_vec<ND>(c, vc);
  g2 aq, bq, aqn, bqn;
  const g2 na = fk2(c_f.negalpha);
  g2_mul_fq(aq, *Qx, a);
  g2_mul_fq(bq, *Qy, b);
  g2_mul(aqn, aq, na);
  g2_mul(bqn, bq, na);
  fl<ND> cl;
  to_limbs<ND>(cl, c);
  g2l Aq, Aqn, Bq, Bqn;
  g2l_make(Aq, aq);
  g2l_make(Aqn, aqn);
  g2l_make(Bq, bq);
  g2l_make(Bqn, bqn);
  f12_stage(v);
#pragma nounroll
  for (int i = 0; i < 6; i++) {
    int j = i + 2, k = i + 3;          // j = i - 4 mod 6, k = i - 3 mod 6
    bool wj = true, wk = true;         // wrapped (needs negalpha) unless i >= 4 / i >= 3
    if (j >= 6) { j -= 6; wj = false; }
    if (k >= 6) { k -= 6; wk = false; }
    g2l fa, fb;
    g2l_sel(fa, Aq, Aqn, wj);
    g2l_sel(fb, Bq, Bqn, wk);
    const fl<ND> vix = ldsf_get(i, 0), viy = ldsf_get(i, 1), vjx = ldsf_get(j, 0), vjy = ldsf_get(j, 1),
                 vkx = ldsf_get(k, 0), vky = ldsf_get(k, 1);
    fl<ND> t;
    g2 o;
    {
      const fl<ND> x[5] = {cl, fa.x, fa.by, fb.x, fb.by}, y[5] = {vix, vjx, vjy, vkx, vky};
      sop_limbs<ND, 5>(t, x, y);
      from_limbs<ND>(o.x, t);
    }
    {
      const fl<ND> x[5] = {cl, fa.x, fa.y, fb.x, fb.y}, y[5] = {viy, vjy, vjx, vky, vkx};
      sop_limbs<ND, 5>(t, x, y);
      from_limbs<ND>(o.y, t);
    }
    v->c[i] = o;                       // v itself is no longer read: it sits in LDS
  }
}

// ---- the Miller accumulator lives in LDS ----------------------------------------------------------------------
// Inside the Miller loop (5-word fields) the accumulator v never touches private memory: it stays in LDS in limb form
// (values below 2q, no conversions between operations), in one of two areas; each operation reads its operands from
// the current area at LDS speed -- also the partner coefficient of a product, whose index depends on the rolled output
// loop -- and writes the result coefficients into the other area.  Register needs stay small (no spills), at the price
// of 72 KB of LDS per workgroup: one wave per SIMD.
static constexpr bool kLdsMiller = FL <= 6;       // (the 8-word fields on this path, one area at one wave per SIMD: 1.25 M against 1.37 M pairings/s)
static constexpr bool kOneArea = kF12Bufs<ND> == 1;
static constexpr bool kPrefetch = kOneArea && PBC_F_PREFETCH != 0;
static PBC_DEV int next_area(int cur) { return kOneArea ? 0 : cur ^ 1; }
// Where the coefficients of a result go while the operand area is still being read.  Two areas: straight into the
// other one.  One area (the default: 36 KB of LDS per workgroup, so that two waves share a SIMD -- a single wave gets a
// multiply-add through only every 9.1 cycles, two share the pipe at 4.6): a 72-word buffer in the lane's private memory,
// copied over the operand once the last coefficient is done (72 scratch stores, 72 loads and 72 LDS stores against the
// ~7000 instructions of an F_q^12 operation).
struct OutArea {
  uint32_t buf[kOneArea ? 12 * FL : 1];
  int dst;
  PBC_DEV void put(int c, int part, const fl<ND> &a) {
    if constexpr (kOneArea) {
#pragma unroll
      for (int l = 0; l < FL; l++) buf[(c * 2 + part) * FL + l] = a.l[l];
    } else {
      ldsf_put(c, part, a, dst);
    }
  }
  // The buffered coefficients back into registers BEFORE the last coefficients of an operation are computed: the loads'
  // latency (the buffers of 2048 resident waves do not fit the L2) passes under a few hundred multiply-adds instead of
  // stalling the wave at the end of the operation (PBC_F_PREFETCH 1; the scheduling barrier keeps the loads where they are)
  template <int NW>
  PBC_DEV void prefetch(uint32_t (&pre)[NW]) {
    if constexpr (kOneArea) {
#pragma unroll
      for (int w = 0; w < NW; w++) pre[w] = buf[w];
#ifndef PBC_HOSTSIM
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  }
  template <int NW>
  PBC_DEV void finish_from(const uint32_t (&pre)[NW]) {
    if constexpr (kOneArea) {
#ifndef PBC_HOSTSIM
      __builtin_amdgcn_sched_barrier(0);         // (or the stores -- and with them the wait for the loads -- are scheduled up into the computation)
#endif
#pragma unroll
      for (int c = 0; c < NW / FL; c++) {
        fl<ND> t;
#pragma unroll
        for (int l = 0; l < FL; l++) t.l[l] = pre[c * FL + l];
        ldsf_put(c >> 1, c & 1, t, 0);
      }
    }
  }
  // only the first `ncoef` coefficients went through the buffer (the sparse line product writes the rest in place)
  PBC_DEV void finish(int ncoef = 6) {
    if constexpr (kOneArea) {
#pragma unroll
      for (int c = 0; c < 12; c++) {
        if (c >= 2 * ncoef) break;
        fl<ND> t;
#pragma unroll
        for (int l = 0; l < FL; l++) t.l[l] = buf[c * FL + l];
        ldsf_put(c >> 1, c & 1, t, 0);
      }
    }
  }
};
static PBC_DEV fl<ND> flk(const uint32_t *w) { fl<ND> r; to_limbs<ND>(r, dk(w)); return r; }   // a constant's limb form
static PBC_DEV void wide_carry(wide<ND> &W, const fl<ND> &oneL) {
  fl<ND> t;
  wide_reduce<ND>(t, W);
  wide_zero<ND>(W);
  wide_mac<ND>(W, t, oneL);
}
template <int BIT>
static PBC_DEV void wide_guard(wide<ND> &Wx, wide<ND> &Wy) {
  if constexpr ((PBC_F_SQZ & BIT) != 0) { wide_squeeze<ND>(Wx); wide_squeeze<ND>(Wy); }
  else { const fl<ND> oneL = fl29(c_f.one29); wide_carry(Wx, oneL); wide_carry(Wy, oneL); }
}
// (accumulator overflow guard of the sums below, four times per squaring / product: wide_squeeze -- the high bits of the
// middle columns move two columns up -- or wide_carry: the running sum goes through a Montgomery reduction and a product
// with R mod q, 2 x 72 multiply-adds; see PBC_F_SQZ)
// area `cur` squared into area 1 - cur.  Coefficient k + 6 of the plain square is reduced first and enters coefficient k
// through X^(6+k) = negalpha X^k as four more products of the same lazy sums: one reduction per output component.
static __device__ __noinline__ void f12_sqr_lds(int cur) {
  fair_tick();
  OutArea O;
  fl<ND> hx, hy;                                 // one area: coefficient 4 waits in registers, 0-3 in the buffer, 5 needs neither
  O.dst = kOneArea ? 0 : 1 - cur;
  // beta y_i: kept for all six coefficients (the compiler puts the array into private memory and reads it back by
  // address: 2-3 scratch loads per pair), or -- PBC_F_BY 1, i-basis only, where it is a negation and a carry pass --
  // recomputed from y_i when a pair needs it
  constexpr bool kByRecompute = kByRe<1>;
  fl<ND> by[kByRecompute ? 1 : 6];
  if constexpr (!kByRecompute) {
#pragma unroll
    for (int i = 0; i < 6; i++) mul_beta(by[i], ldsf_get(i, 1, cur));
  }
#pragma nounroll
  for (int kk = 0; kk < (kPrefetch ? 5 : 6); kk++) {
    fl<ND> t6x, t6y;
    wide<ND> Wx, Wy;
    int units = 0;
#pragma nounroll
    for (int h = 1; h >= 0; h--) {
      if (h && kk == 5) continue;                // X^11 does not occur
      const int k = kk + 6 * h;
      if (!XS || h || kk == 5) {                 // (sparse xi: the wrapped pairs go into the same sums as the plain ones)
        wide_zero<ND>(Wx);
        wide_zero<ND>(Wy);
        units = 0;
      }
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int j = k - i;
        if (j < i || j > 5) continue;            // pairs i <= j (wave-uniform)
        if (units + 4 > kCap) { wide_guard<1>(Wx, Wy); units = 1; }
        const fl<ND> ax = ldsf_get(i, 0, cur), ay = ldsf_get(i, 1, cur);
        fl<ND> byi;
        if constexpr (kByRecompute) mul_beta(byi, ay); else byi = by[i];
        if (XS && h) {
          // a wrapped pair, i + j = kk + 6: a_i a_j X'^6 = a_i (xi' a_j) -- xi' a_j by shifts and adds, then a plain pair
          fl<ND> bx, by2;
          mul_xi(bx, by2, ldsf_get(j, 0, cur), ldsf_get(j, 1, cur));
          if (i != j) { limbs_dbl<ND>(bx, bx); limbs_dbl<ND>(by2, by2); }
          wide_mac<ND>(Wx, ax, bx);
          wide_mac<ND>(Wy, ay, bx);
          wide_mac<ND>(Wx, byi, by2);
          wide_mac<ND>(Wy, ax, by2);
          units += i != j ? 4 : 2;
        } else if (i == j) {                     // a_i^2: re = x^2 + (beta y) y, im = 2 x y
          fl<ND> ax2;
          limbs_dbl<ND>(ax2, ax);
          wide_mac<ND>(Wx, ax, ax);
          wide_mac<ND>(Wx, byi, ay);
          wide_mac<ND>(Wy, ax2, ay);
          units += 2;
        } else {                                 // 2 a_i a_j
          fl<ND> b2;
          limbs_dbl<ND>(b2, ldsf_get(j, 0, cur));
          wide_mac<ND>(Wx, ax, b2);
          wide_mac<ND>(Wy, ay, b2);
          limbs_dbl<ND>(b2, ldsf_get(j, 1, cur));
          wide_mac<ND>(Wx, byi, b2);
          wide_mac<ND>(Wy, ax, b2);
          units += 4;
        }
      }
      if (XS && h) continue;                     // (the plain pairs of coefficient kk follow in the same accumulators)
      if (h) {
        wide_reduce<ND>(t6x, Wx);
        wide_reduce<ND>(t6y, Wy);
      } else {
        if (!XS && kk < 5) {                     // + negalpha * (coefficient k + 6)
          if (units + 2 > kCap) wide_guard<1>(Wx, Wy);
          const fl<ND> nax = fl29(c_f.na29[0]), nay = fl29(c_f.na29[1]), bnay = fl29(c_f.bna29);
          wide_mac<ND>(Wx, nax, t6x);
          wide_mac<ND>(Wx, bnay, t6y);
          wide_mac<ND>(Wy, nax, t6y);
          wide_mac<ND>(Wy, nay, t6x);
        }
        fl<ND> o;                                // (one area: the last coefficient is complete only when every read of the
        wide_reduce<ND>(o, Wx);                  // operand is done -- it goes straight to its slot)
        if (kOneArea && kk == 5) ldsf_put(kk, 0, o, 0); else if (kOneArea && kk == 4) hx = o; else O.put(kk, 0, o);
        wide_reduce<ND>(o, Wy);
        if (kOneArea && kk == 5) ldsf_put(kk, 1, o, 0); else if (kOneArea && kk == 4) hy = o; else O.put(kk, 1, o);
      }
    }
  }
  if constexpr (kPrefetch) {
    // coefficient 5 apart (the cross pairs (0,5), (1,4), (2,3); no fold), with the buffered coefficients 0-3 on their way
    // back into registers meanwhile
    uint32_t pre[8 * FL];
    O.prefetch(pre);
    wide<ND> Wx, Wy;
    wide_zero<ND>(Wx);
    wide_zero<ND>(Wy);
    int units = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (units + 4 > kCap) { wide_guard<1>(Wx, Wy); units = 1; }
      const fl<ND> ax = ldsf_get(i, 0, cur), ay = ldsf_get(i, 1, cur);
      fl<ND> byi, b2;
      if constexpr (kByRecompute) mul_beta(byi, ay); else byi = by[i];
      limbs_dbl<ND>(b2, ldsf_get(5 - i, 0, cur));
      wide_mac<ND>(Wx, ax, b2);
      wide_mac<ND>(Wy, ay, b2);
      limbs_dbl<ND>(b2, ldsf_get(5 - i, 1, cur));
      wide_mac<ND>(Wx, byi, b2);
      wide_mac<ND>(Wy, ax, b2);
      units += 4;
    }
    fl<ND> ox, oy;
    wide_reduce<ND>(ox, Wx);
    wide_reduce<ND>(oy, Wy);
    ldsf_put(5, 0, ox, 0);                       // every read of the operand is done
    ldsf_put(5, 1, oy, 0);
    O.finish_from(pre);
  } else {
    O.finish(4);
  }
  if constexpr (kOneArea) { ldsf_put(4, 0, hx, 0); ldsf_put(4, 1, hy, 0); }
}
// area `cur` times (a Qx X^4 + b Qy X^3 + c) into area 1 - cur (f_miller_evalfn, f_param.c:109-149): the formulas of
// f_line_mul
// (the line's coefficients arrive as limbs: a with limbs up to 2^30 and a value below 2.001 q, b a Montgomery product's
// result, c with limbs <= 2^29 + 6 and a value below 6 q -- what the limb-form steps f_dbl_core_l / f_add_core_l deliver;
// the word-form steps convert theirs)
typedef uint32_t vfl __attribute__((ext_vector_type(FL)));
static PBC_DEV vfl to_vfl(const fl<ND> &a) { vfl v; for (int i = 0; i < FL; i++) v[i] = a.l[i]; return v; }
static PBC_DEV vfl fq_vfl(const fq &a) { fl<ND> t; to_limbs<ND>(t, a); return to_vfl(t); }
static __device__ __noinline__ void f_line_mul_lds(int cur, vfl va, vfl vb, vfl vc, const g2 *Qx, const g2 *Qy) {
  fair_tick();
  OutArea O;
  O.dst = kOneArea ? 0 : 1 - cur;
  fl<ND> al, bl, cl;
#pragma unroll
  for (int i = 0; i < FL; i++) { al.l[i] = va[i]; bl.l[i] = vb[i]; cl.l[i] = vc[i]; }
  g2l Aq, Aqn, Bq, Bqn;
  {
    // a Qx, b Qy and their multiples by -alpha without leaving limb form: four F_q products and two lazily reduced
    // F_q^2 products by the constant (its limbs are scalar operands)
    fl<ND> qx, qy;
    to_limbs<ND>(qx, Qx->x);
    to_limbs<ND>(qy, Qx->y);
    g2l_scale<1>(Aq, qx, qy, al);
    g2l_mul_na(Aqn, Aq);
    to_limbs<ND>(qx, Qy->x);
    to_limbs<ND>(qy, Qy->y);
    g2l_scale(Bq, qx, qy, bl);
    g2l_mul_na(Bqn, Bq);
  }
  uint32_t pre[kPrefetch ? 6 * FL : 1];
#if PBC_F_LINE_SEL
  // the factors of an output change twice in the six iterations (b Qy loses its xi at output 3, a Qx at output 4): two
  // branches on the loop counter that copy 18 registers when taken, instead of 36 selects in every iteration
  g2l fa = Aqn, fb = Bqn;
#endif
#pragma nounroll
  for (int half = 0; half < 2; half++) {
    if constexpr (kPrefetch) { if (half == 1) O.prefetch(pre); }      // outputs 0-2 come back while 3-5 are computed
#pragma nounroll
    for (int i = 3 * half; i < 3 * half + 3; i++) {
      int j = i + 2, k = i + 3;          // j = i - 4 mod 6, k = i - 3 mod 6
      bool wj = true, wk = true;         // wrapped (needs negalpha) unless i >= 4 / i >= 3
      if (j >= 6) { j -= 6; wj = false; }
      if (k >= 6) { k -= 6; wk = false; }
#if PBC_F_LINE_SEL
      (void) wj; (void) wk;
      if (i == 3) { fb = Bq; PBC_KEEP_BRANCH(); }
      if (i == 4) { fa = Aq; PBC_KEEP_BRANCH(); }
#else
      g2l fa, fb;
      g2l_sel(fa, Aq, Aqn, wj);
      g2l_sel(fb, Bq, Bqn, wk);
#endif
      const fl<ND> vix = ldsf_get(i, 0, cur), viy = ldsf_get(i, 1, cur), vjx = ldsf_get(j, 0, cur), vjy = ldsf_get(j, 1, cur),
                   vkx = ldsf_get(k, 0, cur), vky = ldsf_get(k, 1, cur);
      fl<ND> t;
      {
        const fl<ND> x[5] = {cl, fa.x, fa.by, fb.x, fb.by}, y[5] = {vix, vjx, vjy, vkx, vky};
        sop_limbs<ND, 5>(t, x, y);
        if (kOneArea && i >= 3) ldsf_put(i, 0, t, 0); else O.put(i, 0, t);
      }
      {
        const fl<ND> x[5] = {cl, fa.x, fa.y, fb.x, fb.y}, y[5] = {viy, vjy, vjx, vky, vkx};
        sop_limbs<ND, 5>(t, x, y);
        if (kOneArea && i >= 3) ldsf_put(i, 1, t, 0); else O.put(i, 1, t);
      }
    }
  }
  // (one area: coefficient s is read by the outputs s, s - 2 and s - 3 mod 6 only, and every operand of an output is in
  // registers before its first component is stored -- outputs 3, 4, 5 overwrite their own slots, 0, 1, 2 wait in the buffer)
  if constexpr (kPrefetch) O.finish_from(pre); else O.finish(3);
}
// area `cur` times the private-memory element b into area 1 - cur (b's limb forms in registers with compile-time
// indices, the accumulator's coefficients from LDS; fold as in f12_sqr_lds)
static __device__ __noinline__ void f12_mul_lds(int cur, const f12 *b) {
  fair_tick();
  OutArea O;
  fl<ND> hx, hy;
  O.dst = kOneArea ? 0 : 1 - cur;
  f12r_t<!kByRe<2>> B;
  f12_load_regs(B, b, false);
#pragma nounroll
  for (int kk = 0; kk < 6; kk++) {
    fl<ND> t6x, t6y;
    wide<ND> Wx, Wy;
    int units = 0;
#pragma nounroll
    for (int h = 1; h >= 0; h--) {
      if (h && kk == 5) continue;
      const int k = kk + 6 * h;
      if (!XS || h || kk == 5) {                 // (sparse xi: the wrapped pairs go into the same sums as the plain ones)
        wide_zero<ND>(Wx);
        wide_zero<ND>(Wy);
        units = 0;
      }
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int j = k - i;
        if (j < 0 || j > 5) continue;            // wave-uniform
        if (units + 2 > kCap) { wide_guard<2>(Wx, Wy); units = 1; }
        fl<ND> ax = ldsf_get(j, 0, cur), ay = ldsf_get(j, 1, cur);
        if (XS && h) mul_xi(ax, ay, ax, ay);     // b_i a_j X'^6 = b_i (xi' a_j)
        wide_mac<ND>(Wx, B.x[i], ax);
        wide_mac<ND>(Wx, f12r_by(B, i), ay);
        wide_mac<ND>(Wy, B.x[i], ay);
        wide_mac<ND>(Wy, B.y[i], ax);
        units += 2;
      }
      if (XS && h) continue;
      if (h) {
        wide_reduce<ND>(t6x, Wx);
        wide_reduce<ND>(t6y, Wy);
      } else {
        if (!XS && kk < 5) {
          if (units + 2 > kCap) wide_guard<2>(Wx, Wy);
          const fl<ND> nax = fl29(c_f.na29[0]), nay = fl29(c_f.na29[1]), bnay = fl29(c_f.bna29);
          wide_mac<ND>(Wx, nax, t6x);
          wide_mac<ND>(Wx, bnay, t6y);
          wide_mac<ND>(Wy, nax, t6y);
          wide_mac<ND>(Wy, nay, t6x);
        }
        fl<ND> o;                                // (one area: the last coefficient is complete only when every read of the
        wide_reduce<ND>(o, Wx);                  // operand is done -- it goes straight to its slot)
        if (kOneArea && kk == 5) ldsf_put(kk, 0, o, 0); else if (kOneArea && kk == 4) hx = o; else O.put(kk, 0, o);
        wide_reduce<ND>(o, Wy);
        if (kOneArea && kk == 5) ldsf_put(kk, 1, o, 0); else if (kOneArea && kk == 4) hy = o; else O.put(kk, 1, o);
      }
    }
  }
  O.finish(4);
  if constexpr (kOneArea) { ldsf_put(4, 0, hx, 0); ldsf_put(4, 1, hy, 0); }
}
// ---- squaring in the cyclotomic subgroup (Granger-Scott) ------------------------------------------------------------------
// After the easy part of the final exponentiation an element a = sum c_i X^i has order dividing q^4 - q^2 + 1.  With
// W = X^3 (W^2 = xi = negalpha), F_q^12 = F_q^4[X]/(X^3 - W), a = g0 + g1 X + g2 X^2, g_j = c_j + c_(j+3) W, and
//     a^2 = (3 g0^2 - 2 conj g0) + (3 W g2^2 + 2 conj g1) X + (3 g1^2 - 2 conj g2) X^2        (conj: W -> -W),
// coefficient by coefficient (checked against the plain square on f.param before it was written, and by every vector):
//     c0' = F(c0, c3, c0)   c3' = G(c0, c3, c3)        F(u, v, l) = 3 u^2 + 3 xi v^2 - 2 l
//     c2' = F(c1, c4, c2)   c5' = G(c1, c4, c5)        G(u, v, l) = 6 u v + 2 l
//     c4' = F(c2, c5, c4)   c1' = H(c2, c5, c1)        H(u, v, l) = 6 xi u v + 2 l
// -- about 75 product / reduction units against the 114 of the plain square (132 with a general beta), nothing to buffer
// (a pair's results overwrite its own operands), every output one lazily reduced sum.  Small multiples ride on the
// operands (3 u_x has limbs below 3 2^29 and counts three units of a column's nine), the linear terms enter as products
// with R mod q, differences through the borrowed constant K = 8 q.
typedef uint32_t vcyc __attribute__((ext_vector_type(4 * FL)));
static PBC_DEV void lnorm(fl<ND> &r, const uint32_t *t) {       // parallel carry pass: limbs below 2^32 in, <= 2^29 + 8 out
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < FL; i++) {
    r.l[i] = (i < FL - 1 ? (t[i] & Limbs29<ND>::MASK) : t[i]) + c;
    c = t[i] >> 29;
  }
}
static PBC_DEV void lscale(fl<ND> &r, const fl<ND> &a, uint32_t k) {
#pragma unroll
  for (int i = 0; i < FL; i++) r.l[i] = a.l[i] * k;
}
// (F, G) or (F, H) of the pair u = c_iu, v = c_iv with the linear terms c_ilf, c_ilg (coefficients of the LDS area)
static __device__ __noinline__ vcyc cyc_pair(int iu_, int iv_, int ilf_, int ilg_, int with_xi_) {
#ifdef PBC_HOSTSIM
  const int iu = iu_, iv = iv_, ilf = ilf_, ilg = ilg_, with_xi = with_xi_;
#else
  const int iu = __builtin_amdgcn_readfirstlane(iu_), iv = __builtin_amdgcn_readfirstlane(iv_), ilf = __builtin_amdgcn_readfirstlane(ilf_),
            ilg = __builtin_amdgcn_readfirstlane(ilg_), with_xi = __builtin_amdgcn_readfirstlane(with_xi_);
#endif
  const fl<ND> ux = ldsf_get(iu, 0), uy = ldsf_get(iu, 1), vx = ldsf_get(iv, 0), vy = ldsf_get(iv, 1);
  const fl<ND> oneL = fl29(c_f.one29);
  fl<ND> bvy, Vx, Vy, t, u3x, buy3, uy2, kx, ky, Fx, Fy, Gx, Gy;
  mul_beta(bvy, vy);
  {                                              // V = v^2
    const fl<ND> x[2] = {vx, bvy}, y[2] = {vx, vy};
    sop_limbs<ND, 2>(Vx, x, y);
    limbs_dbl<ND>(t, vx);
    const fl<ND> x1[1] = {t}, y1[1] = {vy};
    sop_limbs<ND, 1, 1>(Vy, x1, y1);
  }
  {                                              // K - 2 l for the linear term of F
    uint32_t d[FL];
    const fl<ND> lx = ldsf_get(ilf, 0), ly = ldsf_get(ilf, 1);
#pragma unroll
    for (int i = 0; i < FL; i++) d[i] = c_f.kneg8_29[i] - 2 * lx.l[i];
    lnorm(kx, d);
#pragma unroll
    for (int i = 0; i < FL; i++) d[i] = c_f.kneg8_29[i] - 2 * ly.l[i];
    lnorm(ky, d);
  }
  lscale(u3x, ux, 3);
  mul_beta(t, uy);
  lscale(buy3, t, 3);
  limbs_dbl<ND>(uy2, uy);
  if constexpr (XS) {                            // F = 3 u^2 + (3 xi' V - 2 l): xi' V by shifts and adds, the linear part as ONE product with R mod q
    fl<ND> Tx, Ty, linx, liny;
    uint32_t d[FL];
    mul_xi(Tx, Ty, Vx, Vy);                      // limbs <= 2^29 + 8, values below 16 q
#pragma unroll
    for (int i = 0; i < FL; i++) d[i] = 3 * Tx.l[i] + kx.l[i];      // kx = K - 2 l.x: limbs <= 2^29 + 8
    lnorm(linx, d);
#pragma unroll
    for (int i = 0; i < FL; i++) d[i] = 3 * Ty.l[i] + ky.l[i];
    lnorm(liny, d);
    const fl<ND> x[3] = {u3x, buy3, linx}, y[3] = {ux, uy, oneL};
    sop_limbs<ND, 3, 4>(Fx, x, y);               // 3 + 3 + 1 units
    const fl<ND> x1[2] = {u3x, liny}, y1[2] = {uy2, oneL};
    sop_limbs<ND, 2, 5>(Fy, x1, y1);             // 6 + 1 units
  } else {                                       // F = 3 u^2 + 3 xi V - 2 l
    const fl<ND> x[5] = {u3x, buy3, fl29(c_f.cyc29[0]), fl29(c_f.cyc29[2]), kx}, y[5] = {ux, uy, Vx, Vy, oneL};
    sop_limbs<ND, 5, 4>(Fx, x, y);               // 3 + 3 + 1 + 1 + 1 units
    const fl<ND> x1[4] = {u3x, fl29(c_f.cyc29[0]), fl29(c_f.cyc29[1]), ky}, y1[4] = {uy2, Vy, Vx, oneL};
    sop_limbs<ND, 4, 5>(Fy, x1, y1);             // 6 + 1 + 1 + 1 units
  }
  fl<ND> l2x, l2y;
  {
    const fl<ND> lx = ldsf_get(ilg, 0), ly = ldsf_get(ilg, 1);
    limbs_dbl<ND>(l2x, lx);
    limbs_dbl<ND>(l2y, ly);
  }
  if (with_xi) {                                 // H = 6 xi (u v) + 2 l: the product reduced first
    fl<ND> Tx, Ty;
    {
      const fl<ND> x[2] = {ux, uy}, y[2] = {vx, bvy};
      sop_limbs<ND, 2>(Tx, x, y);
      const fl<ND> x1[2] = {ux, uy}, y1[2] = {vy, vx};
      sop_limbs<ND, 2>(Ty, x1, y1);
    }
    if constexpr (XS) {                          // 6 xi' T + 2 l: one product with R mod q per component
      fl<ND> Xx, Xy, lin;
      uint32_t d[FL];
      mul_xi(Xx, Xy, Tx, Ty);
#pragma unroll
      for (int i = 0; i < FL; i++) d[i] = 6 * Xx.l[i] + l2x.l[i];
      lnorm(lin, d);
      { const fl<ND> x[1] = {lin}, y[1] = {oneL}; sop_limbs<ND, 1, 1>(Gx, x, y); }
#pragma unroll
      for (int i = 0; i < FL; i++) d[i] = 6 * Xy.l[i] + l2y.l[i];
      lnorm(lin, d);
      { const fl<ND> x[1] = {lin}, y[1] = {oneL}; sop_limbs<ND, 1, 1>(Gy, x, y); }
    } else {
    const fl<ND> x[3] = {fl29(c_f.cyc29[3]), fl29(c_f.cyc29[5]), l2x}, y[3] = {Tx, Ty, oneL};
    sop_limbs<ND, 3, 1>(Gx, x, y);
    const fl<ND> x1[3] = {fl29(c_f.cyc29[3]), fl29(c_f.cyc29[4]), l2y}, y1[3] = {Ty, Tx, oneL};
    sop_limbs<ND, 3, 1>(Gy, x1, y1);
    }
  } else {                                       // G = 6 u v + 2 l
    fl<ND> wx, wy;
    uint32_t d[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) d[i] = 6 * ux.l[i];
    lnorm(wx, d);
#pragma unroll
    for (int i = 0; i < FL; i++) d[i] = 6 * uy.l[i];
    lnorm(wy, d);
    const fl<ND> x[3] = {wx, wy, l2x}, y[3] = {vx, bvy, oneL};
    sop_limbs<ND, 3, 1>(Gx, x, y);
    const fl<ND> x1[3] = {wx, wy, l2y}, y1[3] = {vy, vx, oneL};
    sop_limbs<ND, 3, 1>(Gy, x1, y1);
  }
  vcyc r;
#pragma unroll
  for (int i = 0; i < FL; i++) { r[i] = Fx.l[i]; r[FL + i] = Fy.l[i]; r[2 * FL + i] = Gx.l[i]; r[3 * FL + i] = Gy.l[i]; }
  return r;
}
static PBC_DEV void cyc_store(vcyc r, int cf, int cg) {
  fl<ND> t;
#pragma unroll
  for (int i = 0; i < FL; i++) t.l[i] = r[i];
  ldsf_put(cf, 0, t);
#pragma unroll
  for (int i = 0; i < FL; i++) t.l[i] = r[FL + i];
  ldsf_put(cf, 1, t);
#pragma unroll
  for (int i = 0; i < FL; i++) t.l[i] = r[2 * FL + i];
  ldsf_put(cg, 0, t);
#pragma unroll
  for (int i = 0; i < FL; i++) t.l[i] = r[3 * FL + i];
  ldsf_put(cg, 1, t);
}
// area 0 squared in place (one-area layout; the element must lie in the cyclotomic subgroup)
static __device__ __noinline__ void f12_cyc_sqr_lds() {
  fair_tick();
  const vcyc r0 = cyc_pair(0, 3, 0, 3, 0);
  cyc_store(r0, 0, 3);                           // c0, c3 are read by no other pair
  const vcyc r1 = cyc_pair(1, 4, 2, 5, 0);       // c2', c5'
  const vcyc r2 = cyc_pair(2, 5, 4, 1, 1);       // c4', c1' -- reads the old c2, c5, c4, c1
  cyc_store(r1, 2, 5);
  cyc_store(r2, 4, 1);
}

// a private-memory element into area `buf`
static PBC_DEV void f12_lds_import(const f12 *a, int buf) {
#pragma nounroll
  for (int c = 0; c < 6; c++) {
    fl<ND> x, y;
    to_limbs<ND>(x, a->c[c].x);
    to_limbs<ND>(y, a->c[c].y);
    ldsf_put(c, 0, x, buf);
    ldsf_put(c, 1, y, buf);
  }
}
// 1 into area `buf`; area `buf` out to a private-memory element (canonical words)
static PBC_DEV void f12_lds_one(int buf) {
  fl<ND> oneL, z;
  to_limbs<ND>(oneL, dk(fpk<ND>().one));
#pragma unroll
  for (int l = 0; l < FL; l++) z.l[l] = 0;
#pragma nounroll
  for (int c = 0; c < 6; c++) { ldsf_put(c, 0, c == 0 ? oneL : z, buf); ldsf_put(c, 1, z, buf); }
}
static PBC_DEV void f12_lds_export(f12 *r, int buf) {
#pragma nounroll
  for (int c = 0; c < 6; c++) {
    g2 t;
    from_limbs<ND>(t.x, ldsf_get(c, 0, buf));
    from_limbs<ND>(t.y, ldsf_get(c, 1, buf));
    r->c[c] = t;
  }
}

// ---- point arithmetic on E(F_q): y^2 = x^3 + b in limb form (six-limb fields) -----------------------------------------
// The Miller loop's tangent / chord steps with V = (X, Y, Z) and the fixed P kept as 6-limb elements in the redundant
// representation of pairing_al.cuh / pairing_d.cuh (no conditional subtractions; differences through borrowed multiples
// of q; a parallel carry pass where a product needs normalised limbs), so that a step is ~20 limb products without a
// conversion -- the word-form steps spend 110 instructions around the 72 multiply-adds of every product.  The formulas
// and the classes (limb size u in units of 2^29, value bound B in units of q) are those of d_dbl_core_l / d_add_core_l
// with a = 0; the host mirror tracks and asserts them.
static constexpr bool kLimbPoint = FL == 6 && kLdsMiller;
typedef fl<ND> el;
enum { PK2 = 0, PK4 = 1, PK16 = 2, PK32 = 3 };                // (c, D) = (2, 1), (4, 2), (16, 2), (32, 2)
#ifdef PBC_HOSTSIM
static constexpr double PU_STRICT = 1.0 - 1.0 / 536870912.0, PU_ALMOST = 1.0 + 7.0 / 536870912.0;
static constexpr double PKC[4] = {2, 4, 16, 32}, PKD[4] = {1, 2, 2, 2};
static constexpr double PSLACK = 16384.0;                     // R / q > 2^14
static void ph_fail(const char *what, double v) { fprintf(stderr, "hostsim: limb-form type f point arithmetic: %s (%g)\n", what, v); abort(); }
static void ph_limbs(const el &a) {
  for (int i = 0; i < FL; i++)
    if ((double) a.l[i] > a.hs_u * 536870912.0) ph_fail("limb above its tracked bound", a.hs_u);
}
static void ph_set(el &r, double u, double B) { r.hs_u = u; r.hs_B = B; ph_limbs(r); }
static void ph_dom(const el &b, int k) {
  ph_limbs(b);
  if (b.hs_u > PKD[k] * PU_STRICT + 1e-12) ph_fail("subtrahend limbs not dominated", b.hs_u);
  // top limb: K's is at least c q / 2^145 - 1 - D, b's at most B q / 2^145, and q >= 2^152 (init checks it: pl_ok)
  if ((PKC[k] - b.hs_B) * 128.0 < PKD[k] + 1) ph_fail("subtrahend value not dominated", b.hs_B);
}
static void ph_cols(double s) { if (s > (64.0 - FL) / FL - 0.01) ph_fail("column capacity", s); }
#define PL_HS(...) __VA_ARGS__
#else
#define PL_HS(...)
#endif
static PBC_DEV el pl_const(int k) {
  el r;
#pragma unroll
  for (int l = 0; l < FL; l++) r.l[l] = c_f.pk29[k][l < 6 ? l : 5];
  return r;
}
static PBC_DEV void pl_from_fq(el &r, const fq &a) { to_limbs<ND>(r, a); PL_HS(ph_set(r, PU_STRICT, 1.0);) }
static PBC_DEV void pl_add(el &r, const el &a, const el &b) {
#pragma unroll
  for (int i = 0; i < FL; i++) r.l[i] = a.l[i] + b.l[i];
  PL_HS(if (a.hs_u + b.hs_u >= 8) ph_fail("sum overflows 32 bits", a.hs_u + b.hs_u); ph_set(r, a.hs_u + b.hs_u, a.hs_B + b.hs_B);)
}
template <int S>
static PBC_DEV void pl_shl(el &r, const el &a) {
#pragma unroll
  for (int i = 0; i < FL; i++) r.l[i] = a.l[i] << S;
  PL_HS(if (a.hs_u * (1 << S) >= 8) ph_fail("shift overflows 32 bits", a.hs_u); ph_set(r, a.hs_u * (1 << S), a.hs_B * (1 << S));)
}
static PBC_DEV void pl_subk(el &r, const el &a, const el &b, int k) {      // a + K_k - b
  const el K = pl_const(k);
  PL_HS(ph_dom(b, k); const double u = a.hs_u + PKD[k] + 1, B = a.hs_B + PKC[k]; if (u >= 8) ph_fail("difference overflows 32 bits", u);)
#pragma unroll
  for (int i = 0; i < FL; i++) r.l[i] = a.l[i] - b.l[i] + K.l[i];
  PL_HS(ph_set(r, u, B);)
}
static PBC_DEV void pl_negk(el &r, const el &b, int k) {
  const el K = pl_const(k);
  PL_HS(ph_dom(b, k);)
#pragma unroll
  for (int i = 0; i < FL; i++) r.l[i] = K.l[i] - b.l[i];
  PL_HS(ph_set(r, PKD[k] + 1, PKC[k]);)
}
static PBC_DEV void pl_norm(el &r, const el &a) {             // parallel carry pass: limbs <= 2^29 + 6
  uint32_t c = 0;
  PL_HS(ph_limbs(a); const double B = a.hs_B; if (a.hs_u >= 8) ph_fail("normalising limbs above 32 bits", a.hs_u);)
#pragma unroll
  for (int i = 0; i < FL; i++) {
    const uint32_t t = a.l[i];
    r.l[i] = (i < FL - 1 ? (t & Limbs29<ND>::MASK) : t) + c;
    c = t >> 29;
  }
  PL_HS(ph_set(r, PU_ALMOST, B);)
}
template <int UNITS>                                          // UNITS: the product of the operands' limb sizes (a column holds 9)
static PBC_DEV void pl_mul(el &r, const el &a, const el &b) {
  const el x[1] = {a}, y[1] = {b};
  PL_HS(ph_limbs(a); ph_limbs(b); ph_cols(a.hs_u * b.hs_u); if (a.hs_u * b.hs_u > UNITS + 0.001) ph_fail("more product units than declared", a.hs_u * b.hs_u);
        const double B = 1 + a.hs_B * b.hs_B / PSLACK;)
  sop_limbs<ND, 1, UNITS - 1>(r, x, y);
  PL_HS(ph_set(r, PU_STRICT, B);)
}
static PBC_DEV void pl_sqr(el &r, const el &a) {              // a (almost) normalised
  PL_HS(const el x[1] = {a}; ph_limbs(a); if (a.hs_u > PU_ALMOST) ph_fail("squaring an unnormalised element", a.hs_u); hs_sop_check<ND>(x, x, 1);
        const double B = 1 + a.hs_B * a.hs_B / PSLACK;)
  sqr_limbs<ND>(r.l, a.l);
  PL_HS(ph_set(r, PU_STRICT, B);)
}
static PBC_DEV void pl_sop2(el &r, const el &a0, const el &b0, const el &a1, const el &b1) {   // a0 b0 + a1 b1, one reduction
  const el x[2] = {a0, a1}, y[2] = {b0, b1};
  PL_HS(ph_limbs(a0); ph_limbs(b0); ph_limbs(a1); ph_limbs(b1); ph_cols(a0.hs_u * b0.hs_u + a1.hs_u * b1.hs_u);
        if (a0.hs_u * b0.hs_u + a1.hs_u * b1.hs_u > 2.01) ph_fail("two-term sum of unnormalised operands", a0.hs_u);
        const double B = 1 + (a0.hs_B * b0.hs_B + a1.hs_B * b1.hs_B) / PSLACK;)
  sop_limbs<ND, 2, 0>(r, x, y);
  PL_HS(ph_set(r, PU_STRICT, B);)
}
struct pjac { el X, Y, Z; };           // X, Y almost normalised with B <= 18; Z = 2 Y Z: limbs up to 2^30, B 3
// tangent at V and V <- 2V: a' = -M Z^2 (u 2), b' = (2YZ) Z^2 (P-class), c' = M X - 2Y^2 (almost normalised); M = 3X^2
static PBC_DEV void f_dbl_core_l(pjac &V, el &la, el &lb, el &lc) {
  el ZZ, XX, YY, M, t0, t1, S1, Z3, W, Xn, Yn;
  pl_mul<4>(ZZ, V.Z, V.Z);
  pl_sqr(XX, V.X);
  pl_sqr(YY, V.Y);
  pl_shl<1>(M, XX);
  pl_add(M, M, XX);                    // u 3, B 4
  pl_norm(M, M);
  pl_mul<1>(la, M, ZZ);
  pl_negk(la, la, PK2);                // u 2, B 2
  pl_mul<2>(Z3, V.Y, V.Z);
  pl_shl<1>(Z3, Z3);                   // 2YZ: u 2, B 3
  pl_mul<2>(lb, Z3, ZZ);
  pl_mul<1>(lc, M, V.X);
  pl_shl<1>(t1, YY);                   // u 2, B 3
  pl_subk(lc, lc, t1, PK4);            // u 4, B 5.5
  pl_norm(lc, lc);
  pl_mul<1>(S1, V.X, YY);              // X Y^2
  pl_shl<3>(t1, S1);                   // 8 X Y^2: u < 8, B 12
  pl_norm(t1, t1);
  pl_sqr(t0, M);
  pl_subk(t0, t0, t1, PK16);           // X3 = M^2 - 2S, S = 4 X Y^2: u 4, B 17.5
  pl_norm(Xn, t0);
  pl_shl<2>(t1, S1);                   // S: u 4, B 6
  pl_subk(W, t1, Xn, PK32);            // S - X3: u 7, B 38
  pl_mul<7>(t0, M, W);
  pl_sqr(t1, YY);
  pl_shl<3>(t1, t1);                   // 8 Y^4: u < 8, B 12
  pl_norm(t1, t1);
  pl_subk(t0, t0, t1, PK16);           // Y3 = M (S - X3) - 8Y^4: u 4, B 17.5
  pl_norm(Yn, t0);
  V.X = Xn;
  V.Y = Yn;
  V.Z = Z3;
}
// chord through V and +-P, V <- V +- P: a' = -R = Y - Py Z^3, b' = Z3, c' = R Px - Z3 Py   (Py: the sign already applied)
static PBC_DEV void f_add_core_l(pjac &V, const el &Px, const el &Py, el &la, el &lb, el &lc) {
  el ZZ, H, Rn, HH, HHH, t0, t1, Z3, W, Xn, nY;
  pl_mul<4>(ZZ, V.Z, V.Z);
  pl_mul<1>(H, Px, ZZ);
  pl_subk(H, H, V.X, PK32);            // u 4, B 33.5
  pl_norm(H, H);
  pl_mul<2>(t0, V.Z, ZZ);
  pl_mul<1>(t0, Py, t0);
  pl_subk(Rn, V.Y, t0, PK2);           // -R: u 4, B 19.5
  pl_norm(Rn, Rn);
  pl_mul<2>(Z3, V.Z, H);
  la = Rn;
  lb = Z3;
  pl_sop2(t0, Rn, Px, Z3, Py);         // -(c'): Rn Px + Z3 Py with Rn = -R
  pl_negk(lc, t0, PK2);                // u 2, B 2
  pl_norm(lc, lc);
  pl_sqr(HH, H);
  pl_mul<1>(HHH, HH, H);
  pl_mul<1>(t0, V.X, HH);              // X1 H^2
  pl_sqr(t1, Rn);
  pl_subk(t1, t1, HHH, PK2);           // u 3, B 3.5
  pl_shl<1>(W, t0);                    // u 2, B 3
  pl_subk(t1, t1, W, PK4);             // X3 = R^2 - H^3 - 2 X1 H^2: u 6, B 7.5
  pl_norm(Xn, t1);
  pl_subk(W, Xn, t0, PK2);             // X3 - X1 H^2: u 3, B 9.5
  pl_norm(W, W);
  pl_negk(nY, V.Y, PK32);              // -Y1: u 3, B 32
  pl_norm(nY, nY);
  pl_sop2(t0, Rn, W, nY, HHH);         // Y3 = R (X1 H^2 - X3) - Y1 H^3 = Rn (X3 - X1 H^2) + (-Y1) H^3
  V.X = Xn;
  V.Y = t0;
  V.Z = Z3;
}

// Miller function: G1 bytes x||y (2 x 20), G2 bytes x||y over F_q^2 (2 x 40)
static __device__ __noinline__ bool f_miller_lane(f12 *v, const uint8_t *g1, const uint8_t *g2b) {
  const int NB = fb();
  fq Px, Py, one;
  g2 Qx, Qy;
  fp_set<ND>(one, fpk<ND>().one);
  fp_load_be<ND>(Px, g1);
  fp_load_be<ND>(Py, g1 + NB);
  g2_load_be(Qx, g2b);
  g2_load_be(Qy, g2b + 2 * NB);
  if constexpr (BM1) {                 // into the i-basis (init_stage3)
    fp_mul<ND>(Qx.y, Qx.y, dk(c_f.cmap));
    fp_mul<ND>(Qy.y, Qy.y, dk(c_f.cmap));
  }
  bool valid;
  {
    // curve_is_valid_point (curve.c:57-77): E: y^2 = x^3 + b;  E': y^2 = x^3 - alpha b over F_q^2
    fq t0, t1;
    fp_sqr<ND>(t0, Px);
    fp_mul<ND>(t0, t0, Px);
    fp_add<ND>(t0, t0, dk(c_f.B));
    fp_sqr<ND>(t1, Py);
    valid = fp_eq<ND>(t0, t1);
    g2 u0, u1;
    g2_sqr(u0, Qx);
    g2_mul(u0, u0, Qx);
    g2_add(u0, u0, fk2(c_f.tb));
    g2_sqr(u1, Qy);
    valid &= g2_eq(u0, u1);
  }
  // untwist: (x, y) -> (x negalphainv X^4, y negalphainv X^3)  (f_pairing, f_param.c:296-303); in the basis X' = X / c of
  // a sparse xi (init_stage4): X^4 / xi = c^-2 X'^4 / xi', X^3 / xi = c^-3 X'^3 / xi'
  {
    const g2 ni = fk2(c_f.negalphainv);
    g2_mul(Qx, Qx, ni);
    g2_mul(Qy, Qy, ni);
    if constexpr (BM1) {
      if (c_f.xs_ok) {
        const g2 ci = fk2(c_f.xc_inv);
        g2 c2;
        g2_sqr(c2, ci);
        g2_mul(Qx, Qx, c2);
        g2_mul(c2, c2, ci);
        g2_mul(Qy, Qy, c2);
      }
    }
  }
  djac V;
  V.X = Px; V.Y = Py; V.Z = one; V.ZZ = one;
  int cur = 0;                         // LDS area holding the accumulator (kLdsMiller)
  if constexpr (kLdsMiller) f12_lds_one(cur); else f12_one(v);
  bool limb = false;                   // the steps on E(F_q) in limb form (wave-uniform: a property of q)
  if constexpr (kLimbPoint) limb = c_f.pl_ok != 0;
  if (limb) {
    if constexpr (kLimbPoint) {
      pjac W;
      el PxL, PyL;
      pl_from_fq(PxL, Px);
      pl_from_fq(PyL, Py);
      W.X = PxL; W.Y = PyL;
      pl_from_fq(W.Z, one);
      // cc_miller_no_denom (f_param.c:216-233): tangent; [double; line+add]; square
      for (int m = c_f.rbits - 2;; m--) {
        {
          el la, lb, lc;
          f_dbl_core_l(W, la, lb, lc);
          f_line_mul_lds(cur, to_vfl(la), to_vfl(lb), to_vfl(lc), &Qx, &Qy);
          cur = next_area(cur);
        }
        if (m <= 0) break;
        const int dig = (int) ((c_f.r[m >> 5] >> (m & 31)) & 1) - (int) ((c_f.rm[m >> 5] >> (m & 31)) & 1);
        if (dig) {                     // chord through V and +-P: the sign is the signed digit of the loop
          el la, lb, lc, Pys = PyL;
          if (dig < 0) {
            pl_negk(Pys, Pys, PK2);
            pl_norm(Pys, Pys);         // B 2
          }
          f_add_core_l(W, PxL, Pys, la, lb, lc);
          f_line_mul_lds(cur, to_vfl(la), to_vfl(lb), to_vfl(lc), &Qx, &Qy);
          cur = next_area(cur);
        }
        f12_sqr_lds(cur);
        cur = next_area(cur);
      }
    }
  } else
  // cc_miller_no_denom (f_param.c:216-233): tangent; [double; line+add]; square
  for (int m = c_f.rbits - 2;; m--) {
    {
      // tangent (do_tangent :171-184, a = 0), scaled by Z^6: M = 3X^2,
      //   a' = -M Z^2, b' = (2YZ) Z^2, c' = M X - 2Y^2;  then V <- 2V
      fq XX, YY, M, t0, t1, S, Z3, la, lb, lc;
      fp_sqr<ND>(XX, V.X);
      fp_sqr<ND>(YY, V.Y);
      fp_dbl<ND>(M, XX);
      fp_add<ND>(M, M, XX);
      fp_mul<ND>(la, M, V.ZZ);
      fp_neg<ND>(la, la);
      fp_mul<ND>(Z3, V.Y, V.Z);
      fp_dbl<ND>(Z3, Z3);
      fp_mul<ND>(lb, Z3, V.ZZ);
      fp_mul<ND>(lc, M, V.X);
      fp_dbl<ND>(t1, YY);
      fp_sub<ND>(lc, lc, t1);
      if constexpr (kLdsMiller) { f_line_mul_lds(cur, fq_vfl(la), fq_vfl(lb), fq_vfl(lc), &Qx, &Qy); cur = next_area(cur); }
      else f_line_mul(v, to_vec<ND>(la), to_vec<ND>(lb), to_vec<ND>(lc), &Qx, &Qy);
      fp_mul<ND>(S, V.X, YY);
      fp_dbl<ND>(S, S);
      fp_dbl<ND>(S, S);
      fp_sqr<ND>(t0, YY);
      fp_dbl<ND>(t0, t0);
      fp_dbl<ND>(t0, t0);
      fp_dbl<ND>(t0, t0);
      fp_sqr<ND>(V.X, M);
      fp_dbl<ND>(t1, S);
      fp_sub<ND>(V.X, V.X, t1);
      fp_sub<ND>(t1, S, V.X);
      fp_mul<ND>(t1, M, t1);
      fp_sub<ND>(V.Y, t1, t0);
      V.Z = Z3;
      fp_sqr<ND>(V.ZZ, Z3);
    }
    if (m <= 0) break;
    const int dig = (int) ((c_f.r[m >> 5] >> (m & 31)) & 1) - (int) ((c_f.rm[m >> 5] >> (m & 31)) & 1);
    if (dig) {
      // chord through V and +-P (do_line :190-199), scaled by Z3 = Z H; the sign is the signed digit of the loop:
      //   a' = -R, b' = Z3, c' = R Px - Z3 Py;  V <- V +- P
      fq Pys;
      fp_neg<ND>(Pys, Py);
      fp_cmov<ND>(Pys, Py, dig > 0);
      fq H, R, HH, HHH, t0, t1, Z3, la, lc;
      fp_mul<ND>(H, Px, V.ZZ);
      fp_sub<ND>(H, H, V.X);
      fp_mul<ND>(t0, V.Z, V.ZZ);
      fp_mul<ND>(R, Pys, t0);
      fp_sub<ND>(R, R, V.Y);
      fp_mul<ND>(Z3, V.Z, H);
      fp_neg<ND>(la, R);
      fp_mul<ND>(lc, R, Px);
      fp_mul<ND>(t0, Z3, Pys);
      fp_sub<ND>(lc, lc, t0);
      if constexpr (kLdsMiller) { f_line_mul_lds(cur, fq_vfl(la), fq_vfl(Z3), fq_vfl(lc), &Qx, &Qy); cur = next_area(cur); }
      else f_line_mul(v, to_vec<ND>(la), to_vec<ND>(Z3), to_vec<ND>(lc), &Qx, &Qy);
      fp_sqr<ND>(HH, H);
      fp_mul<ND>(HHH, HH, H);
      fp_mul<ND>(t0, V.X, HH);
      fp_sqr<ND>(t1, R);
      fp_sub<ND>(t1, t1, HHH);
      fp_sub<ND>(t1, t1, t0);
      fp_sub<ND>(t1, t1, t0);
      fp_sub<ND>(t0, t0, t1);
      fp_mul<ND>(t0, R, t0);
      fp_mul<ND>(HHH, V.Y, HHH);
      fp_sub<ND>(V.Y, t0, HHH);
      V.X = t1;
      V.Z = Z3;
      fp_sqr<ND>(V.ZZ, Z3);
    }
    if constexpr (kLdsMiller) { f12_sqr_lds(cur); cur = next_area(cur); } else f12_sqr(v, v);
  }
  if constexpr (kLdsMiller) f12_lds_export(v, cur);
  return valid;
}

// a^q: conjugate the F_q^2 coefficients and scale by gamma^i (X^q = gamma X)
static __device__ __noinline__ void f12_frob(f12 *r, const f12 *a) {
  const g2 gm = fk2(c_f.gamma);
  g2 gpow = gm;
  {
    g2 t = a->c[0];
    fp_neg<ND>(t.y, t.y);
    r->c[0] = t;
  }
#pragma nounroll
  for (int i = 1; i < 6; i++) {
    g2 t = a->c[i], u;
    fp_neg<ND>(t.y, t.y);
    g2_mul(u, t, gpow);
    r->c[i] = u;
    g2_mul(gpow, gpow, gm);
  }
}
// a^|x| by square-and-multiply (wave-uniform bits); on the 5-word fields the running power stays in LDS
static __device__ __noinline__ void f12_pow_x(f12 *r, const f12 *a) {
  if constexpr (kLdsMiller) {
    int cur = 0;
    f12_lds_import(a, cur);
#pragma nounroll
    for (int i = c_f.bn_xbits - 2; i >= 0; i--) {
      // (every caller's `a` is a power of the easy part's result: cyclotomic subgroup)
      bool done = false;
      if constexpr (kOneArea && kCap >= 9) {     // (the sums of cyc_pair take nine product units: six-limb fields)
        if (c_f.cyc_ok) { f12_cyc_sqr_lds(); done = true; }
      }
      if (!done) { f12_sqr_lds(cur); cur = next_area(cur); }
      if ((c_f.bn_x[i >> 5] >> (i & 31)) & 1) { f12_mul_lds(cur, a); cur = next_area(cur); }
    }
    f12_lds_export(r, cur);                      // r may be a: a is no longer read
  } else {
    f12 acc = *a;
#pragma nounroll
    for (int i = c_f.bn_xbits - 2; i >= 0; i--) {
      f12_sqr(&acc, &acc);
      if ((c_f.bn_x[i >> 5] >> (i & 31)) & 1) f12_mul(&acc, &acc, a);
    }
    *r = acc;
  }
}

// Hard part out^((q^4-q^2+1)/r) for BN parameters.  With l3 = 1, l2 = 6x^2+1,
// l1 = -36x^3-18x^2-12x+1, l0 = -36x^3-30x^2-18x-2 the exponent equals
// l0 + l1 q + l2 q^2 + l3 q^3 (checked by the host for the actual q, r), evaluated with the
// vector chain  y0 y1^2 y2^6 y3^12 y4^18 y5^30 y6^36  (Scott et al., "On the final
// exponentiation for calculating pairings on ordinary elliptic curves").  After the easy
// part `out` is in the cyclotomic subgroup, where inversion is the q^6 Frobenius.
// The reference raises to the same integer with generic_pow_mpz (field.c:14-126).
// The y_i are formed one at a time and multiplied into the two running products at once: seven F_q^12 temporaries
// instead of thirteen in the lane's private memory (8.4 -> 3.4 KB per lane for the whole kernel).
static __device__ __noinline__ void f_hard_bn(f12 *out) {
  f12 fx, fx2, fx3, t0, t1, y, u;
  f12_pow_x(&fx, out);
  if (c_f.bn_xneg) f12_conj(&fx, &fx);
  f12_pow_x(&fx2, &fx);
  if (c_f.bn_xneg) f12_conj(&fx2, &fx2);
  f12_pow_x(&fx3, &fx2);
  if (c_f.bn_xneg) f12_conj(&fx3, &fx3);
  // T0 = y6^2 y4 y5;  T1 = y3 y5 T0;  T0 = T0 y2;  T1 = (T1^2 T0)^2;  T0 = T1 y1;  T1 = T1 y0;  out = T0^2 T1
  // y6 = 1 / (f^(x^3) (f^(x^3))^q)
  f12_frob(&u, &fx3);
  f12_mul(&y, &u, &fx3);
  f12_conj(&y, &y);
  f12_sqr(&t0, &y);
  // y4 = 1 / (f^x (f^(x^2))^q)
  f12_frob(&u, &fx2);
  f12_mul(&y, &u, &fx);
  f12_conj(&y, &y);
  f12_mul(&t0, &t0, &y);
  // y5 = 1 / f^(x^2)
  f12_conj(&y, &fx2);
  f12_mul(&t0, &t0, &y);
  // y3 = 1 / (f^x)^q
  f12_frob(&u, &fx);
  f12_conj(&u, &u);
  f12_mul(&t1, &u, &y);
  f12_mul(&t1, &t1, &t0);
  // y2 = (f^(x^2))^(q^2)
  f12_qpower(&y, &fx2, c_f.xpowq2);
  f12_mul(&t0, &t0, &y);
  f12_sqr(&t1, &t1);
  f12_mul(&t1, &t1, &t0);
  f12_sqr(&t1, &t1);
  // y1 = 1/f
  f12_conj(&y, out);
  f12_mul(&t0, &t1, &y);
  // y0 = f^q f^(q^2) f^(q^3)
  f12_frob(&u, out);
  f12_qpower(&y, out, c_f.xpowq2);
  f12_mul(&fx, &u, &y);
  f12_frob(&u, &y);
  f12_mul(&fx, &fx, &u);
  f12_mul(&t1, &t1, &fx);
  f12_sqr(&t0, &t0);
  f12_mul(out, &t0, &t1);
}
// generic parameters: element_pow_mpz(out, out, tateexp).  generic_pow_mpz (field.c:14-126) slides a window; any addition
// chain gives the same group element, and this one is plain square-and-multiply: no table in the lane's private memory
// (pbc_param_init_f_gen only produces BN parameters, which take f_hard_bn).
static __device__ __noinline__ void f_hard_generic(f12 *out) {
  f12 acc = *out;
#pragma nounroll
  for (int i = c_f.tebits - 2; i >= 0; i--) {
    f12_sqr(&acc, &acc);
    if ((c_f.tateexp[i >> 5] >> (i & 31)) & 1) f12_mul(&acc, &acc, out);
  }
  *out = acc;
}
// f_tateexp (f_param.c:250-283)
static __device__ __noinline__ void f_final_exp(f12 *out) {
  {
    f12 x, y;
    f12_qpower(&y, out, c_f.xpowq8);
    f12_conj(&x, out);
    f12_mul(&y, &y, &x);
    f12_qpower(&x, out, c_f.xpowq2);
    f12_mul(&x, &x, out);
    f12_inv(&x, &x);
    f12_mul(out, &y, &x);
  }
  if (c_f.bn_ok) f_hard_bn(out);
  else f_hard_generic(out);
}

// element_pairing (f_pairing) / element_prod_pairing (generic_prod_pairings, ecc/pairing.c:35-46:
// Type F installs no dedicated product routine; the product of k reduced pairings equals the
// reduced product of the Miller functions) for one lane
// miller_only: diagnostic (stop before the final exponentiation)
static __device__ void f_prod_pairing_lane(uint8_t *gt, const uint8_t *g1, const uint8_t *g2b, int k,
                                    bool miller_only = false) {
  f12 F;
  bool valid = f_miller_lane(&F, g1, g2b);
  for (int j = 1; j < k; j++) {
    f12 f;
    valid &= f_miller_lane(&f, g1 + (size_t) j * 2 * fb(), g2b + (size_t) j * 4 * fb());
    f12_mul(&F, &F, &f);
  }
#ifdef PBC_F_WHATIF_MILLER_ONLY
  miller_only = true;                            // what-if timing only (tools/whatif_time.py): wrong results
#endif
  if (!miller_only) f_final_exp(&F);
  if (!valid) f12_one(&F);
  g2 cpw;                                        // c^-i: back from the basis X' = X / c of a sparse xi (init_stage4)
  g2_zero(cpw);
  fp_set<ND>(cpw.x, fpk<ND>().one);
#pragma nounroll
  for (int i = 0; i < 6; i++) {
    g2 o = F.c[i];
    if constexpr (BM1) {
      if (c_f.xs_ok && valid && !miller_only) {
        if (i) { g2_mul(cpw, cpw, fk2(c_f.xc_inv)); g2_mul(o, o, cpw); }
      }
      fp_mul<ND>(o.y, o.y, dk(c_f.cinv));        // back to the reference's basis of F_q^2
    }
    g2_store_be(gt + 2 * fb() * i, o);
  }
}

// bring-up diagnostics: one tower primitive on operands given as GT-format bytes
static __device__ void f_debug_lane(int op, uint8_t *out, const uint8_t *inA, const uint8_t *inB) {
  f12 A, B, R;
#pragma nounroll
  for (int i = 0; i < 6; i++) { g2_load_be(A.c[i], inA + 2 * fb() * i); g2_load_be(B.c[i], inB + 2 * fb() * i); }
  if (op == 10) f12_mul(&R, &A, &B);
  else if (op == 11) f12_sqr(&R, &A);
  else if (op == 12) { R = A; f_line_mul(&R, to_vec<ND>(B.c[0].x), to_vec<ND>(B.c[0].y), to_vec<ND>(B.c[3].x), &B.c[1], &B.c[2]); }
  else if (op == 13) f12_qpower(&R, &A, c_f.xpowq2);
  else if (op == 14) f12_inv(&R, &A);
  else if (op == 15) { R = A; f12_sqr(&R, &R); f12_mul(&R, &R, &B); }
  else { R = A; f_final_exp(&R); }
#pragma nounroll
  for (int i = 0; i < 6; i++) g2_store_be(out + 2 * fb() * i, R.c[i]);
}

// ---- device-side derivation of the tower constants ---------------------------------------
// stage 1: F_q-level constants and negalpha (beta must be in c_f before any g2_mul)
static PBC_DEV void init_stage1(FConst *out, const FRaw &raw, const FConst &base) {
  FConst C = base;
  fq r2, t, b, be, a0, a1;
  fp_set<ND>(r2, fpk<ND>().r2);
  fp_set<ND>(t, raw.b); fp_mul<ND>(b, t, r2);
  fp_set<ND>(t, raw.beta); fp_mul<ND>(be, t, r2);
  fp_set<ND>(t, raw.alpha0); fp_mul<ND>(a0, t, r2); fp_neg<ND>(a0, a0);
  fp_set<ND>(t, raw.alpha1); fp_mul<ND>(a1, t, r2); fp_neg<ND>(a1, a1);
  for (int k = 0; k < ND; k++) {
    C.B[k] = b.v[k]; C.beta[k] = be.v[k];
    C.negalpha[0][k] = a0.v[k]; C.negalpha[1][k] = a1.v[k];
  }
  {
    fq one;
    fp_set<ND>(one, fpk<ND>().one);
    fl<ND> bel, o, nx, ny, bny;
    to_limbs<ND>(bel, be);
    to_limbs<ND>(o, one);
    to_limbs<ND>(nx, a0);
    to_limbs<ND>(ny, a1);
    { const fl<ND> xx[1] = {ny}, yy[1] = {bel}; sop_limbs<ND, 1>(bny, xx, yy); }
    for (int l = 0; l < Limbs29<ND>::L; l++) {
      C.beta29[l] = bel.l[l]; C.one29[l] = o.l[l]; C.na29[0][l] = nx.l[l]; C.na29[1][l] = ny.l[l]; C.bna29[l] = bny.l[l];
    }
  }
  *out = C;
}
// the constants of the cyclotomic squaring from C.negalpha and C.beta (either basis)
static PBC_DEV void fill_cyc(FConst &C, const FRaw &raw) {
  fq nx, ny, bny, acc[3];
  fp_set<ND>(nx, C.negalpha[0]);
  fp_set<ND>(ny, C.negalpha[1]);
  fp_mul<ND>(bny, ny, dk(C.beta));
  const fq base[3] = {nx, ny, bny};
  for (int k = 0; k < 3; k++) {
    fp_dbl<ND>(acc[k], base[k]);
    fp_add<ND>(acc[k], acc[k], base[k]);           // 3 x
    fl<ND> l;
    to_limbs<ND>(l, acc[k]);
    for (int i = 0; i < FL; i++) C.cyc29[k][i] = l.l[i];
    fp_dbl<ND>(acc[k], acc[k]);                    // 6 x
    to_limbs<ND>(l, acc[k]);
    for (int i = 0; i < FL; i++) C.cyc29[3 + k][i] = l.l[i];
  }
  for (int i = 0; i < 9; i++) { C.kneg29[i] = raw.kneg29[i]; C.kneg8_29[i] = raw.kneg8_29[i]; }
  C.cyc_ok = raw.k_ok;
  for (int t = 0; t < 4; t++) for (int i = 0; i < 6; i++) C.pk29[t][i] = raw.pk29[t][i];
  C.pl_ok = raw.pl_ok;
}
// stage 2 (c_f holds stage 1): negalphainv, twist b, and the Frobenius constants:
//   X^q = negalpha^((q-1)/6) X =: c X,  X^(q^2) = conj(c) c X = N(c) X,  X^(q^6) = N(c)^3 X,
//   X^(q^8) = N(c)^4 X   (the reference gets the same values by brute-force powering,
//   f_param.c:431-444)
static PBC_DEV void init_stage2(FConst *out, const FRaw &raw) {
  FConst C = c_f;
  const g2 na = fk2(c_f.negalpha);
  g2 ni, tb, c;
  g2_inv(ni, na);
  g2_mul_fq(tb, na, dk(c_f.B));
  fq one;
  fp_set<ND>(one, fpk<ND>().one);
  g2_zero(c);
  c.x = one;
  for (int i = raw.e6bits - 1; i >= 0; i--) {
    g2_sqr(c, c);
    if ((raw.e6[i >> 5] >> (i & 31)) & 1) g2_mul(c, c, na);
  }
  fq n, n2, n3, n4, t;
  fp_sqr<ND>(n, c.x);
  fp_sqr<ND>(t, c.y);
  fp_mul<ND>(t, t, dk(c_f.beta));
  fp_sub<ND>(n, n, t);                 // N(c) = c conj(c)
  fp_sqr<ND>(n2, n);
  fp_mul<ND>(n3, n2, n);
  fp_sqr<ND>(n4, n2);
  for (int k = 0; k < ND; k++) {
    C.negalphainv[0][k] = ni.x.v[k]; C.negalphainv[1][k] = ni.y.v[k];
    C.tb[0][k] = tb.x.v[k]; C.tb[1][k] = tb.y.v[k];
    C.xpowq2[0][k] = n.v[k];  C.xpowq2[1][k] = 0;
    C.xpowq6[0][k] = n3.v[k]; C.xpowq6[1][k] = 0;
    C.xpowq8[0][k] = n4.v[k]; C.xpowq8[1][k] = 0;
    C.gamma[0][k] = c.x.v[k]; C.gamma[1][k] = c.y.v[k];
  }
  fill_cyc(C, raw);
  *out = C;
}

// stage 3 (c_f holds stage 2; q = 3 mod 4, raw.e4bits > 0): the i-basis copy of the constants for the pairing kernels.
// beta is a non-residue and so is -1, hence -beta = c^2 with c = (-beta)^((q+1)/4); s -> c i maps F_q[s]/(s^2 - beta)
// onto F_q[i]/(i^2 + 1), (x, y) -> (x, c y), fixing F_q: every F_q^2 constant of the tower gets its second component
// scaled by c, "beta" becomes -1 and products by it turn into negations (mul_beta).  The kernels map Q on the way in
// and the GT coefficients on the way out.  out->bm1 stays 0 when c^2 != -beta (cannot happen for valid parameters).
static PBC_DEV void init_stage3(FConst *out, const FRaw &raw) {
  FConst C = c_f;
  C.bm1 = 0;
  fq one, nb, c, t;
  fp_set<ND>(one, fpk<ND>().one);
  fp_neg<ND>(nb, dk(c_f.beta));
  c = one;
  for (int i = raw.e4bits - 1; i >= 0; i--) {
    fp_sqr<ND>(c, c);
    if ((raw.e4[i >> 5] >> (i & 31)) & 1) fp_mul<ND>(c, c, nb);
  }
  fp_sqr<ND>(t, c);
  if (raw.e4bits > 0 && fp_eq<ND>(t, nb)) {
    fq ci, m1, y;
    fp_inv<ND>(ci, c);
    fp_neg<ND>(m1, one);
    uint32_t (*const f2[4])[NF_MAX] = {C.negalpha, C.negalphainv, C.gamma, C.tb};
    for (int k = 0; k < 4; k++) {
      fp_set<ND>(y, f2[k][1]);
      fp_mul<ND>(y, y, c);
      for (int w = 0; w < ND; w++) f2[k][1][w] = y.v[w];
    }
    fl<ND> l;
    to_limbs<ND>(l, m1);
    for (int w = 0; w < ND; w++) { C.beta[w] = m1.v[w]; C.cmap[w] = c.v[w]; C.cinv[w] = ci.v[w]; }
    for (int i = 0; i < Limbs29<ND>::L; i++) C.beta29[i] = l.l[i];
    fp_set<ND>(y, C.negalpha[1]);
    to_limbs<ND>(l, y);
    for (int i = 0; i < Limbs29<ND>::L; i++) C.na29[1][i] = l.l[i];
    fp_neg<ND>(y, y);                  // beta * negalpha.y = -negalpha.y
    to_limbs<ND>(l, y);
    for (int i = 0; i < Limbs29<ND>::L; i++) C.bna29[i] = l.l[i];
    fill_cyc(C, raw);
    C.bm1 = 1;
  }
  *out = C;
}

// stage 4 (c_f holds the stage-3 i-basis block): look for a small xi' = u + v i, u, v in {1, 2, 4}, in the class of
// xi = negalpha modulo 6th powers -- xi / xi' is a 6th power iff (xi / xi')^((q^2-1)/6) = 1 -- take c with c^6 = xi / xi'
// (the root through the decomposition of F_q^2* into its 2-3-part, generated by xi^m, and the rest; integers from the host)
// and rewrite the block for the basis X' = X / c.  out->xs_ok stays 0 when no candidate fits (or a check fails).
static PBC_DEV void g2_pow(g2 &r, const g2 &a, const uint32_t *e, int bits) {
  fq one;
  fp_set<ND>(one, fpk<ND>().one);
  g2_zero(r);
  r.x = one;
  for (int i = bits - 1; i >= 0; i--) {
    g2_sqr(r, r);
    if ((e[i >> 5] >> (i & 31)) & 1) g2_mul(r, r, a);
  }
}
static PBC_DEV void init_stage4(FConst *out, const FRaw &raw) {
  FConst C = c_f;
  C.xs_ok = 0;
  C.xs_u = C.xs_v = 0;
  fq one;
  fp_set<ND>(one, fpk<ND>().one);
  g2 unit;
  g2_zero(unit);
  unit.x = one;
  for (int k = 0; k < ND; k++) { C.xc_inv[0][k] = one.v[k]; C.xc_inv[1][k] = 0; }
  const g2 xi = fk2(c_f.negalpha);
  bool found = false;
  g2 z, w;
  if (raw.xs_try && c_f.bm1) {
    // (v <= 2: u x - v y is formed with the borrowed constant 8 q, which dominates 2 y)
    static const int cand[6][2] = {{1, 1}, {2, 1}, {1, 2}, {2, 2}, {4, 1}, {4, 2}};
    for (int t = 0; t < 6 && !found; t++) {
      g2_zero(z);
      for (int k = 0; k < cand[t][0]; k++) fp_add<ND>(z.x, z.x, one);
      for (int k = 0; k < cand[t][1]; k++) fp_add<ND>(z.y, z.y, one);
      g2 zi, cls;
      g2_inv(zi, z);
      g2_mul(w, xi, zi);
      g2_pow(cls, w, raw.xs_ecls, raw.xs_eclsbits);
      if (g2_eq(cls, unit)) { found = true; C.xs_u = cand[t][0]; C.xs_v = cand[t][1]; }
    }
  }
  if (found) {
    g2 g, wS, cm, acc = unit, cS = unit, c, chk;
    g2_pow(g, xi, raw.xs_m, raw.xs_mbits);           // generates the subgroup of order S
    g2_pow(wS, w, raw.xs_eS, raw.xs_eSbits);
    g2_pow(cm, w, raw.xs_em, raw.xs_embits);
    int j = -1;
    for (int t = 0; t < raw.xs_S; t++) {
      if (g2_eq(acc, wS)) { j = t; break; }
      g2_mul(acc, acc, g);
    }
    found = j >= 0 && j % 6 == 0;
    for (int t = 0; found && t < j / 6; t++) g2_mul(cS, cS, g);
    g2_mul(c, cm, cS);
    g2_sqr(chk, c);
    g2_mul(chk, chk, c);
    g2_sqr(chk, chk);                                // c^6
    found = found && g2_eq(chk, w);
    if (found) {
      g2 ci, zi, gm, tbv = fk2(c_f.tb);
      g2_inv(ci, c);
      g2_inv(zi, z);
      g2_pow(gm, z, raw.e6, raw.e6bits);             // X'^q = xi'^((q-1)/6) X'
      fq n, n2, n3, n4, t, ny;
      fp_sqr<ND>(n, gm.x);
      fp_sqr<ND>(t, gm.y);
      fp_add<ND>(n, n, t);                           // N(gamma') in the i-basis: x^2 + y^2
      fp_sqr<ND>(n2, n);
      fp_mul<ND>(n3, n2, n);
      fp_sqr<ND>(n4, n2);
      fp_neg<ND>(ny, z.y);
      fl<ND> lx, ly, lby;
      to_limbs<ND>(lx, z.x);
      to_limbs<ND>(ly, z.y);
      to_limbs<ND>(lby, ny);                         // beta * xi'.y = -xi'.y
      for (int k = 0; k < ND; k++) {
        C.negalpha[0][k] = z.x.v[k]; C.negalpha[1][k] = z.y.v[k];
        C.negalphainv[0][k] = zi.x.v[k]; C.negalphainv[1][k] = zi.y.v[k];
        C.gamma[0][k] = gm.x.v[k]; C.gamma[1][k] = gm.y.v[k];
        C.xpowq2[0][k] = n.v[k]; C.xpowq6[0][k] = n3.v[k]; C.xpowq8[0][k] = n4.v[k];
        C.xc_inv[0][k] = ci.x.v[k]; C.xc_inv[1][k] = ci.y.v[k];
        C.tb[0][k] = tbv.x.v[k]; C.tb[1][k] = tbv.y.v[k];
      }
      for (int l = 0; l < FL; l++) { C.na29[0][l] = lx.l[l]; C.na29[1][l] = ly.l[l]; C.bna29[l] = lby.l[l]; }
      fill_cyc(C, raw);
      C.xs_su = C.xs_u == 4 ? 2 : C.xs_u == 2 ? 1 : 0;
      C.xs_sv = C.xs_v == 2 ? 1 : 0;
      C.xs_ok = 1;
    }
  }
  *out = C;
}

};  // struct TypeF

}  // namespace pbc
