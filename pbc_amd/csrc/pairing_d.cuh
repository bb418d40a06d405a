// pairing_d.cuh -- Type D (MNT curve y^2 = x^3 + ax + b over F_q, embedding degree k = 6)
// and Type G (Freeman curve, k = 10) reduced Tate pairings, one pairing per lane.  The two
// families share one construction in the reference (ecc/d_param.c, ecc/g_param.c): with d = k/2,
// F_q^d = F_q[x]/(f), F_q^k = F_q^d[sqrt(v)], G2 on the quadratic twist over F_q^d.  Here they
// are one template over the field width (words) and d; the comments below cite d_param.c, the
// type g twins are cc_miller_no_denom_affine g_param.c:308-408, lucas_even :413-469,
// tatepower10 :471-536, cc_pairing :541-558, g_init_pairing :1248-1354.
//
// Computes the same GT value as the reference's cc_pairing (ecc/d_param.c:570-587):
// twist map, cc_miller_no_denom_affine (:321-422) and cc_tatepower (:505-564, k = 6 branch
// with lucas_even :441-502), but re-derived for a GPU:
//   * towers on fixed-width limbs: F_q^3 = F_q[x]/(x^3 + c2 x^2 + c1 x + c0) (the polymod
//     ring of arith/poly.c with its x^3, x^4 reduction table, :1302-1333) and
//     F_q^6 = F_q^3[sqrt(v)] (generic quadratic extension, arith/fieldquadratic.c fq_*);
//   * the Miller loop runs in Jacobian coordinates on E(F_q): the reference's affine loop
//     costs one F_q inversion per doubling/addition (248 per pairing); the
//     projective line coefficients differ from the affine ones by factors in F_q^*, which
//     (q^6-1)/r kills;
//   * final exponentiation with a single F_q inversion:  m^(q^3-1) = conj(m)^2 / N(m);
//     raising to q+1 keeps numerator w = u^q u and denominator D = N^q N apart; with
//     w = A + B sqrt(v) the Lucas parameter is P = 2A/D and the closing division of
//     lucas_even by P^2 - 4 = 4 v (B/D)^2 folds into D/(v B): one inversion of D*B in
//     F_q^3, done through the norm to F_q.
#pragma once
#include "fp.cuh"

#ifndef PBC_D_FAIR_BIT
#define PBC_D_FAIR_BIT 21
#endif
namespace pbc {
// Resident workgroups + time-sliced priorities (host_common.h, fp.cuh): same-box A/B, ms per 2^18 launch -- round 3: d201
// 41.6 -> 32.8, d190 31.5 -> 58.4, 16-term products of d159 160 -> 185, the type g instantiation faults with the loop around
// its body; d159 single pairings: 14.9 -> 16.2 then, 14.45 -> 14.00 in round 4 (profiles/r04_notes.md) and, unlike the
// plain grid, the same on boxes whose dispatch rounds do not pack (14.5 / 17.5 ms).  A per-instantiation
// choice, and for the 5-word field per launch: single pairings and pairing_pp_apply resident, products on the plain grid
// with the priorities switched off for that launch (pbc_hip_d.hip).
#ifndef PBC_D_RES5
#define PBC_D_RES5 1
#endif
#ifndef PBC_G_RES
#define PBC_G_RES 0                    // experiment switch: the type g instantiation too (tools/g_resident_fault.md)
#endif
// Round 4, the 6-word fields (d190: same-box A/B, ms per 2^18 launch, profiles/r04_ab_d190.txt): fused F_q^6 bodies on
// the plain grid 71.6 (the 36-word accumulator overflows the 32 argument registers: 1 024 spilled registers, and the
// round-4 build runs it at less than half its round-3 rate), Karatsuba over the out-of-line F_q^3 products 31.6, the same
// on resident workgroups 24.6 = 10.6 M pairings/s.  So: the fused bodies for the 5-word fields only, 6 words resident.
#ifndef PBC_D_RES6
#define PBC_D_RES6 1
#endif
#ifndef PBC_D_FUSED_MAXN
#define PBC_D_FUSED_MAXN 5             // widest field whose F_q^6 operations run as one fused limb-form body (kFusedF6)
#endif
template <int N, int DEG> constexpr bool kDResident = ((N == 7 || (N == 5 && PBC_D_RES5) || (N == 6 && PBC_D_RES6)) && DEG == 3) || (PBC_G_RES && N == 5 && DEG == 5);

constexpr int ND_MAX = 7;              // widest MNT field built in: 224-bit q (d224.param)
constexpr int DEG_MAX = 5;             // d = k/2: 3 (type d), 5 (type g)
struct DConst {                        // pptr (ecc/d_param.c:40-51) + curve/field constants
  uint32_t A[ND_MAX], B[ND_MAX];       // curve coefficients (Montgomery form)
  // x^d .. x^(2d-2) mod f (poly.c compute_x_powers) as 29-bit LIMBS, packed [(t d + i) L + limb] for the object's own
  // d and L: the products read them as scalar operands of the multiply-adds, no conversion instructions
  uint32_t xpwr29[120];                // max (d - 1) d L over the built-in (words, d) pairs: 4 * 5 * 6
  uint32_t nqr29[8];                   // v in limbs (f3_mul_v)
  uint32_t nqr[ND_MAX], nqrinv[ND_MAX], nqrinv2[ND_MAX];   // v, v^-1, v^-2 (d_param.c:1028-1032, :1072-1075)
  uint32_t xpowq[DEG_MAX - 1][DEG_MAX][ND_MAX];  // x^q, x^2q, (x^3q, x^4q) (d_param.c:1044-1050, g_param.c:1307-1316)
  uint32_t ta[ND_MAX], tb[ND_MAX];     // twist: y^2 = x^3 + a v^2 x + b v^3 (curve.c:885-901)
  uint32_t r[9], rm[9];                // Miller loop digits: NAF of r >> 1, +1 digits in r[], -1 digits in rm[] (hostbn.h); a 256-bit r can put its leading digit at position 256
  uint32_t phik[16];                   // Phi_k(q)/r (d_param.c:1036-1042, g_param.c:1288-1305)
  int rbits, phikbits;
  // 1: the 5-word d = 3 kernels keep the point state in limb form (kLimbPoint).  Set by the host when q fills at least
  // 8 bits of its top limb (the borrowed subtraction constants take their borrow from that limb) and the parameter text
  // does not say "hip_no_limb 1"; otherwise the same kernels run the word-form step routines.
  int limb_ok;
};
static_assert(sizeof(DConst) <= KOFF_XS - KOFF_TYPE, "constant block layout");
#define c_d (pbc::kconst<pbc::DConst, pbc::KOFF_TYPE>())
struct DRaw { uint32_t a[ND_MAX], b[ND_MAX], coeff[DEG_MAX][ND_MAX], nqr[ND_MAX], q[ND_MAX + 1]; int qbits; };

constexpr int D_LANES = 128;
typedef uint32_t v32 __attribute__((ext_vector_type(32)));
// Per-lane Miller state in LDS, word-major ([word][lane]: conflict-free); one array per instantiation.
// (d = 3 on the 5-word fields keeps the five F_q slots of the point state as 6-limb elements, see kLimbPoint)
template <int ND, int DEG> constexpr int kPointWords = (DEG == 3 && ND == 5) ? Limbs29<ND>::L : ND;
template <int ND, int DEG> __shared__ uint32_t g_lds_d[(2 * DEG * ND + 5 * kPointWords<ND, DEG>) * D_LANES];

// Everything below is per field width and extension degree: ND 32-bit words per F_q element (5 for
// the 159-bit d159 and 149-bit g149 fields, 6 / 7 for the 175..224-bit fields of the other shipped
// type d files), DEG = d.  Names keep the type d flavour: f3 = F_q^d, f6 = F_q^k.
template <int ND, int DEG>
struct TypeMNT {
// time-sliced priorities between the two waves of a SIMD (fp.cuh pbc_fair_tick): for the instantiations that are launched
// with resident workgroups (kDResident)
static PBC_DEV void d_fair_tick() {
  if constexpr (kDResident<ND, DEG>) pbc_fair_tick<PBC_D_FAIR_BIT>();
}
typedef fp<ND> fq;
struct f3 { fq c[DEG]; };              // c0 + c1 x + ... + c(d-1) x^(d-1)
struct f6 { f3 x, y; };                // x + y sqrt(v)

static PBC_DEV fq dk(const uint32_t *w) { fq r; fp_set<ND>(r, w); return r; }

// ---- F_q^3 ------------------------------------------------------------------------------
static PBC_DEV void f3_add(f3 &r, const f3 &a, const f3 &b) { for (int i = 0; i < DEG; i++) fp_add<ND>(r.c[i], a.c[i], b.c[i]); }
static PBC_DEV void f3_sub(f3 &r, const f3 &a, const f3 &b) { for (int i = 0; i < DEG; i++) fp_sub<ND>(r.c[i], a.c[i], b.c[i]); }
static PBC_DEV void f3_dbl(f3 &r, const f3 &a) { for (int i = 0; i < DEG; i++) fp_dbl<ND>(r.c[i], a.c[i]); }
static PBC_DEV void f3_neg(f3 &r, const f3 &a) { for (int i = 0; i < DEG; i++) fp_neg<ND>(r.c[i], a.c[i]); }
static PBC_DEV void f3_halve(f3 &r, const f3 &a) { for (int i = 0; i < DEG; i++) fp_halve<ND>(r.c[i], a.c[i]); }
// polymod_const_mul (poly.c:1550-1558)
static PBC_DEV void f3_mul_fq(f3 &r, const f3 &a, const fq &s) { for (int i = 0; i < DEG; i++) fp_mul<ND>(r.c[i], a.c[i], s); }
static PBC_DEV bool f3_eq(const f3 &a, const f3 &b) {
  int e = 1;
#pragma unroll
  for (int i = 0; i < DEG; i++) e &= (int) fp_eq<ND>(a.c[i], b.c[i]);
  return e != 0;
}
static PBC_DEV void f3_set_fq(f3 &r, const fq &s) {
#pragma unroll
  for (int i = 0; i < DEG; i++)
#pragma unroll
    for (int k = 0; k < ND; k++) r.c[i].v[k] = (i == 0) ? s.v[k] : 0;
}

// F_q^d product with lazy reduction (polymod_mul_degree3 poly.c:910-930 / polymod_mul :880-908: same
// ring element).  With h_t the x^(d+t) coefficient of the plain polynomial product and X_t = x^(d+t) mod f:
//   h_t = sum_{i+j=d+t} a_i b_j                              (reduced once each)
//   c_k = sum_{i+j=k} a_i b_j + sum_t h_t X_t[k]             (one reduction per output coefficient)
// d = 3: 15 limb products + 5 Montgomery reductions instead of 12 full products (6 Karatsuba + 6 table);
// d = 5: 45 limb products + 9 reductions instead of 25 + 20 full products.
static_assert((DEG - 1) * DEG * Limbs29<ND>::L <= 120 && Limbs29<ND>::L <= 8, "DConst limb tables");
static PBC_DEV fl<ND> xpwr_limbs(int t, int i) {
  fl<ND> r;
#pragma unroll
  for (int l = 0; l < Limbs29<ND>::L; l++) r.l[l] = c_d.xpwr29[(t * DEG + i) * Limbs29<ND>::L + l];
  return r;
}
template <int S, bool TABLE>
static PBC_DEV void mul_coeff(fl<ND> &out, const fl<ND> *A, const fl<ND> *B, const fl<ND> *H) {
  constexpr int lo = S - (DEG - 1) > 0 ? S - (DEG - 1) : 0, hi = S < DEG - 1 ? S : DEG - 1;
  constexpr int NDIR = hi - lo + 1, T = NDIR + (TABLE ? DEG - 1 : 0);
  fl<ND> x[T], y[T];
#pragma unroll
  for (int i = lo; i <= hi; i++) { x[i - lo] = A[i]; y[i - lo] = B[S - i]; }
  if constexpr (TABLE) {
#pragma unroll
    for (int t = 0; t < DEG - 1; t++) { x[NDIR + t] = H[t]; y[NDIR + t] = xpwr_limbs(t, S); }
  }
  sop_limbs<ND, T>(out, x, y);
}
// squares: cross terms once with a doubled operand (polymod_square_degree3, poly.c:1049-1089)
template <int S, bool TABLE>
static PBC_DEV void sqr_coeff(fl<ND> &out, const fl<ND> *A, const fl<ND> *A2, const fl<ND> *H) {
  constexpr int lo = S - (DEG - 1) > 0 ? S - (DEG - 1) : 0, hic = (S + 1) / 2 - 1;   // pairs i < S - i
  constexpr int NC = hic - lo + 1 > 0 ? hic - lo + 1 : 0;
  constexpr bool SQ = (S % 2 == 0) && (S / 2 < DEG);
  constexpr int T = NC + (SQ ? 1 : 0) + (TABLE ? DEG - 1 : 0);
  fl<ND> x[T], y[T];
#pragma unroll
  for (int i = lo; i <= hic; i++) { x[i - lo] = A2[i]; y[i - lo] = A[S - i]; }
  if constexpr (SQ) { x[NC] = A[S / 2]; y[NC] = A[S / 2]; }
  if constexpr (TABLE) {
#pragma unroll
    for (int t = 0; t < DEG - 1; t++) { x[NC + SQ + t] = H[t]; y[NC + SQ + t] = xpwr_limbs(t, S); }
  }
  sop_limbs<ND, T, NC>(out, x, y);
}
template <int S> static PBC_DEV void mul_high(fl<ND> *H, const fl<ND> *A, const fl<ND> *B) {
  if constexpr (S < 2 * DEG - 1) { mul_coeff<S, false>(H[S - DEG], A, B, H); mul_high<S + 1>(H, A, B); }
}
template <int S> static PBC_DEV void mul_low(f3 &r, const fl<ND> *A, const fl<ND> *B, const fl<ND> *H) {
  if constexpr (S < DEG) { fl<ND> c; mul_coeff<S, true>(c, A, B, H); from_limbs<ND>(r.c[S], c); mul_low<S + 1>(r, A, B, H); }
}
template <int S> static PBC_DEV void sqr_high(fl<ND> *H, const fl<ND> *A, const fl<ND> *A2) {
  if constexpr (S < 2 * DEG - 1) { sqr_coeff<S, false>(H[S - DEG], A, A2, H); sqr_high<S + 1>(H, A, A2); }
}
template <int S> static PBC_DEV void sqr_low(f3 &r, const fl<ND> *A, const fl<ND> *A2, const fl<ND> *H) {
  if constexpr (S < DEG) { fl<ND> c; sqr_coeff<S, true>(c, A, A2, H); from_limbs<ND>(r.c[S], c); sqr_low<S + 1>(r, A, A2, H); }
}
static PBC_DEV void f3_mul_inl(f3 &r, const f3 &a, const f3 &b) {
  fl<ND> A[DEG], B[DEG], H[DEG - 1];
#pragma unroll
  for (int i = 0; i < DEG; i++) { to_limbs<ND>(A[i], a.c[i]); to_limbs<ND>(B[i], b.c[i]); }
  mul_high<DEG>(H, A, B);
  mul_low<0>(r, A, B, H);
}
static PBC_DEV void f3_sqr_inl(f3 &r, const f3 &a) {
  fl<ND> A[DEG], A2[DEG], H[DEG - 1];
#pragma unroll
  for (int i = 0; i < DEG; i++) { to_limbs<ND>(A[i], a.c[i]); limbs_dbl<ND>(A2[i], A[i]); }
  sqr_high<DEG>(H, A, A2);
  sqr_low<0>(r, A, A2, H);
}
// Out-of-line F_q^d product / square (2 d ND / d ND argument words, the first 32 in VGPRs): one
// body each keeps the Miller and Lucas loops inside the instruction cache.
typedef typename vecN<ND>::type v5;
typedef uint32_t f3ret __attribute__((ext_vector_type(DEG * ND)));   // one vector: stays in VGPRs
static PBC_DEV f3ret f3_pack(const f3 &a) {
  f3ret r;
#pragma unroll
  for (int i = 0; i < DEG; i++)
#pragma unroll
    for (int k = 0; k < ND; k++) r[ND * i + k] = a.c[i].v[k];
  return r;
}
static PBC_DEV void f3_unpack(f3 &a, f3ret r) {
#pragma unroll
  for (int i = 0; i < DEG; i++)
#pragma unroll
    for (int k = 0; k < ND; k++) a.c[i].v[k] = r[ND * i + k];
}
static __device__ __noinline__ f3ret f3_mul_call(f3ret va, f3ret vb) {
  f3 a, b, r;
  f3_unpack(a, va);
  f3_unpack(b, vb);
  f3_mul_inl(r, a, b);
  return f3_pack(r);
}
static __device__ __noinline__ f3ret f3_sqr_call(f3ret va) {
  f3 a, r;
  f3_unpack(a, va);
  f3_sqr_inl(r, a);
  return f3_pack(r);
}
static PBC_DEV void f3_mul(f3 &r, const f3 &a, const f3 &b) { f3_unpack(r, f3_mul_call(f3_pack(a), f3_pack(b))); }
static PBC_DEV void f3_sqr(f3 &r, const f3 &a) { f3_unpack(r, f3_sqr_call(f3_pack(a))); }
// a * v for the constant v = nqr of the quadratic extension (three per F_q^k square / product pair): one
// out-of-line body on limb forms instead of d calls of the generic F_q product
static __device__ __noinline__ f3ret f3_mul_v_call(f3ret va) {
  f3 a, r;
  f3_unpack(a, va);
  fl<ND> V;
#pragma unroll
  for (int l = 0; l < Limbs29<ND>::L; l++) V.l[l] = c_d.nqr29[l];
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    fl<ND> x[1], y[1], c;
    to_limbs<ND>(x[0], a.c[i]);
    y[0] = V;
    sop_limbs<ND, 1>(c, x, y);
    from_limbs<ND>(r.c[i], c);
  }
  return f3_pack(r);
}
// (measured: d159 +2 %; for d = 5 the 25-word argument makes it a loss, so type g keeps the generic calls)
static PBC_DEV void f3_mul_v(f3 &r, const f3 &a) {
  if constexpr (DEG == 3) f3_unpack(r, f3_mul_v_call(f3_pack(a)));
  else f3_mul_fq(r, a, dk(c_d.nqr));
}

// a^q on F_q^d: a0 + sum_j a_j x^(jq) (the qpower macros of cc_tatepower, d_param.c:507-527, and
// tatepower10, g_param.c:486-518)
static PBC_DEV void f3_frob(f3 &r, const f3 &a) {
  f3 res;
  fq t;
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    fp_mul<ND>(res.c[i], a.c[1], dk(c_d.xpowq[0][i]));
#pragma unroll
    for (int j = 2; j < DEG; j++) {
      fp_mul<ND>(t, a.c[j], dk(c_d.xpowq[j - 1][i]));
      fp_add<ND>(res.c[i], res.c[i], t);
    }
  }
  fp_add<ND>(res.c[0], res.c[0], a.c[0]);
  r = res;
}
// a^-1 = a^q a^(q^2) ... a^(q^(d-1)) / N(a), N(a) in F_q  (polymod_invert poly.c:521-536 is a
// polynomial extended Euclid; the inverse is unique)
static PBC_DEV void f3_inv(f3 &r, const f3 &a) {
  f3 t, w, m;
  f3_frob(t, a);
  w = t;
  for (int j = 2; j < DEG; j++) {
    f3_frob(t, t);
    f3_mul(w, w, t);
  }
  f3_mul(m, a, w);                     // the norm: only c[0] is non-zero
  fq n;
  fp_inv<ND>(n, m.c[0]);
  f3_mul_fq(r, w, n);
}

// ---- F_q^6 = F_q^3[sqrt(v)] -------------------------------------------------------------
// fq_mul (fieldquadratic.c:197-233): Karatsuba
static PBC_DEV void f6_mul(f6 &r, const f6 &a, const f6 &b) {
  f3 e0, e1, e2, t;
  f3_add(e0, a.x, a.y);
  f3_add(e1, b.x, b.y);
  f3_mul(e2, e0, e1);
  f3_mul(e0, a.x, b.x);
  f3_mul(e1, a.y, b.y);
  f3_mul_v(t, e1);
  f3_add(r.x, t, e0);
  f3_sub(e2, e2, e0);
  f3_sub(r.y, e2, e1);
}
// fq_square (fieldquadratic.c:249-269); here x^2 + v y^2 = (x + y)(x + v y) - (1 + v) xy
static PBC_DEV void f6_sqr(f6 &r, const f6 &a) {
  f3 t, s, vy, u;
  f3_mul(t, a.x, a.y);
  f3_mul_v(vy, a.y);
  f3_add(s, a.x, a.y);
  f3_add(vy, vy, a.x);
  f3_mul(u, s, vy);
  f3_sub(u, u, t);
  f3_mul_v(s, t);
  f3_sub(r.x, u, s);
  f3_dbl(r.y, t);
}

// ---- Miller loop -------------------------------------------------------------------------
struct djac { fq X, Y, Z, ZZ; };

// Per-lane Miller state in LDS, word-major ([word][lane]: conflict-free): the point V = (X, Y, Z),
// the fixed affine P and the untwisted Q.  The two step routines below take NO register
// arguments: a 160-bit F_q product is only ~80 multiply-adds, so calling it out of line costs
// more than it computes; instead each Miller step is ONE out-of-line body with its ~20 products
// inlined, fed from / writing back to LDS, returning just the line value (30 words).
static constexpr int PW = kPointWords<ND, DEG>;
static constexpr bool kLimbPoint = PW != ND;
enum { DL_QX = 0, DL_QY = DEG * ND, DL_X = 2 * DEG * ND, DL_Y = DL_X + PW, DL_Z = DL_X + 2 * PW, DL_PX = DL_X + 3 * PW, DL_PY = DL_X + 4 * PW };
static PBC_DEV fq dl_get(int w) {
  fq r;
#pragma unroll
  for (int k = 0; k < ND; k++) r.v[k] = g_lds_d<ND, DEG>[(w + k) * D_LANES + threadIdx.x];
  return r;
}
static PBC_DEV void dl_put(int w, const fq &a) {
#pragma unroll
  for (int k = 0; k < ND; k++) g_lds_d<ND, DEG>[(w + k) * D_LANES + threadIdx.x] = a.v[k];
}

// l(Q) = (a Qx + c) + (b Qy) sqrt(v) with a, b, c in F_q (d_miller_evalfn, d_param.c:99-111);
// Q is read from LDS, the result is packed for the return registers.  The 6 ND words of a line
// value fit the 32 return VGPRs only for d = 3, ND = 5; otherwise the first call returns the
// sqrt(v)-free half plus b ((d + 1) ND words) and a second, small out-of-line call the other half.
static constexpr bool kLineOneCall = 2 * DEG * ND <= 32;
static_assert((DEG + 1) * ND <= 32, "line value does not fit the return registers");
static PBC_DEV v32 d_evalfn_pack(const fq &a, const fq &b, const fq &c) {
  v32 r;
#pragma unroll
  for (int k = 0; k < 32; k++) r[k] = 0;
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    fq t;
    fp_mul_inl<ND>(t, dl_get(DL_QX + ND * i), a);
    if (i == 0) fp_add<ND>(t, t, c);
#pragma unroll
    for (int k = 0; k < ND; k++) r[ND * i + k] = t.v[k];
    if constexpr (kLineOneCall) {
      fq u;
      fp_mul_inl<ND>(u, dl_get(DL_QY + ND * i), b);
#pragma unroll
      for (int k = 0; k < ND; k++) r[DEG * ND + ND * i + k] = u.v[k];
    }
  }
  if constexpr (!kLineOneCall) {
#pragma unroll
    for (int k = 0; k < ND; k++) r[DEG * ND + k] = b.v[k];
  }
  return r;
}
static __device__ __noinline__ v32 d_line_y_fn(v5 vb) {
  fq b;
  from_vec<ND>(b, vb);
  v32 r;
#pragma unroll
  for (int k = 0; k < 32; k++) r[k] = 0;
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    fq u;
    fp_mul_inl<ND>(u, dl_get(DL_QY + ND * i), b);
#pragma unroll
    for (int k = 0; k < ND; k++) r[ND * i + k] = u.v[k];
  }
  return r;
}
static PBC_DEV void d_unpack(f6 &e0, v32 r) {
#pragma unroll
  for (int i = 0; i < DEG; i++)
#pragma unroll
    for (int k = 0; k < ND; k++) e0.x.c[i].v[k] = r[ND * i + k];
  if constexpr (kLineOneCall) {
#pragma unroll
    for (int i = 0; i < DEG; i++)
#pragma unroll
      for (int k = 0; k < ND; k++) e0.y.c[i].v[k] = r[DEG * ND + ND * i + k];
  } else {
    fq b;
#pragma unroll
    for (int k = 0; k < ND; k++) b.v[k] = r[DEG * ND + k];
    v32 y = d_line_y_fn(to_vec<ND>(b));
#pragma unroll
    for (int i = 0; i < DEG; i++)
#pragma unroll
      for (int k = 0; k < ND; k++) e0.y.c[i].v[k] = y[ND * i + k];
  }
}
// tangent at V (do_tangent d_param.c:344-362, scaled by Z^6 in F_q^*) and V <- 2V:
//   M = 3X^2 + a Z^4,  a' = -M Z^2,  b' = (2YZ) Z^2,  c' = M X - 2Y^2
static PBC_DEV void d_dbl_core(fq &la, fq &lb, fq &lc) {
  fq X = dl_get(DL_X), Y = dl_get(DL_Y), Z = dl_get(DL_Z);
  fq ZZ, XX, YY, M, t0, t1, S, Z3;
  fp_sqr_inl<ND>(ZZ, Z);
  fp_sqr_inl<ND>(XX, X);
  fp_sqr_inl<ND>(YY, Y);
  fp_sqr_inl<ND>(t0, ZZ);
  fp_mul_inl<ND>(t0, t0, dk(c_d.A));
  fp_dbl<ND>(M, XX);
  fp_add<ND>(M, M, XX);
  fp_add<ND>(M, M, t0);
  fp_mul_inl<ND>(la, M, ZZ);
  fp_neg<ND>(la, la);
  fp_mul_inl<ND>(Z3, Y, Z);
  fp_dbl<ND>(Z3, Z3);
  fp_mul_inl<ND>(lb, Z3, ZZ);
  fp_mul_inl<ND>(lc, M, X);
  fp_dbl<ND>(t1, YY);
  fp_sub<ND>(lc, lc, t1);
  fp_mul_inl<ND>(S, X, YY);
  fp_dbl<ND>(S, S);
  fp_dbl<ND>(S, S);
  fp_sqr_inl<ND>(t0, YY);
  fp_dbl<ND>(t0, t0);
  fp_dbl<ND>(t0, t0);
  fp_dbl<ND>(t0, t0);
  fp_sqr_inl<ND>(X, M);
  fp_dbl<ND>(t1, S);
  fp_sub<ND>(X, X, t1);
  fp_sub<ND>(t1, S, X);
  fp_mul_inl<ND>(t1, M, t1);
  fp_sub<ND>(Y, t1, t0);
  dl_put(DL_X, X);
  dl_put(DL_Y, Y);
  dl_put(DL_Z, Z3);
}
static __device__ __noinline__ v32 d_dbl_line_fn() {
  fq la, lb, lc;
  d_dbl_core(la, lb, lc);
  return d_evalfn_pack(la, lb, lc);
}
// chord through V and the affine P (do_line d_param.c:364-379, scaled by Z3 = Z H):
//   H = Px Z^2 - X, R = Py Z^3 - Y;  a' = -R,  b' = Z3,  c' = R Px - Z3 Py;   V <- V + P
// (neg: the chord through V and -P, V <- V - P: the -1 digits of the signed-digit Miller loop)
static PBC_DEV void d_add_core(fq &la, fq &lb, fq &lc, bool neg) {
  fq X = dl_get(DL_X), Y = dl_get(DL_Y), Z = dl_get(DL_Z), Px = dl_get(DL_PX), Py = dl_get(DL_PY);
  {
    fq nPy;
    fp_neg<ND>(nPy, Py);
    fp_cmov<ND>(Py, nPy, neg);
  }
  fq ZZ, H, R, HH, HHH, t0, t1, Z3;
  fp_sqr_inl<ND>(ZZ, Z);
  fp_mul_inl<ND>(H, Px, ZZ);
  fp_sub<ND>(H, H, X);
  fp_mul_inl<ND>(t0, Z, ZZ);
  fp_mul_inl<ND>(R, Py, t0);
  fp_sub<ND>(R, R, Y);
  fp_mul_inl<ND>(Z3, Z, H);
  fp_neg<ND>(la, R);
  fp_mul_inl<ND>(lc, R, Px);
  fp_mul_inl<ND>(t0, Z3, Py);
  fp_sub<ND>(lc, lc, t0);
  fp_sqr_inl<ND>(HH, H);
  fp_mul_inl<ND>(HHH, HH, H);
  fp_mul_inl<ND>(t0, X, HH);
  fp_sqr_inl<ND>(t1, R);
  fp_sub<ND>(t1, t1, HHH);
  fp_sub<ND>(t1, t1, t0);
  fp_sub<ND>(t1, t1, t0);
  fp_sub<ND>(t0, t0, t1);
  fp_mul_inl<ND>(t0, R, t0);
  fp_mul_inl<ND>(HHH, Y, HHH);
  fp_sub<ND>(Y, t0, HHH);
  dl_put(DL_X, t1);
  dl_put(DL_Y, Y);
  dl_put(DL_Z, Z3);
  lb = Z3;
}
static __device__ __noinline__ v32 d_add_line_fn(int neg) {
  fq la, lb, lc;
  d_add_core(la, lb, lc, neg != 0);
  return d_evalfn_pack(la, lb, lc);
}
// ---- fused F_q^k bodies for the Miller loop (d = 3 on the 5-word fields: d159.param) ------------------------------
// f6_mul / f6_sqr above are Karatsuba over three (two) out-of-line F_q^3 products, each of which converts its operands
// into limbs and its result back into words, with word-form additions in between: of the ~3500 instructions of an
// F_q^6 product about 800 are conversions and additions.  Here the whole F_q^6 operation is ONE body on limb forms:
//     (ax + ay s)(bx + by s) = (ax bx + (v ay) by) + (ax by + ay bx) s,      s^2 = v,
// every output coefficient a single lazy sum (up to 8 of the 9 product units a column holds) over BOTH polynomial
// products, reduced once: 51 limb products + 13 reductions against Karatsuba's 48 + 18, and no additions at all.
// The accumulator travels in registers (30 words in, 30 out); the line functions multiply their value into it directly,
// so the line value never exists in word form.
// (6-word fields: the 36-word accumulator overflows the 32 argument registers by four words, which travel through the
// stack; their 7-limb columns hold 8 product units, so the line value's "+ c'" is normalised instead of counted double)
static constexpr bool kFusedF6 = DEG == 3 && ND <= PBC_D_FUSED_MAXN;
static constexpr bool kLineDbl = Limbs29<ND>::L <= 6;
typedef uint32_t f6vec __attribute__((ext_vector_type(2 * DEG * ND)));
struct f6l { fl<ND> x[DEG], y[DEG]; };
static PBC_DEV f6vec f6_pack(const f6 &a) {
  f6vec r;
#pragma unroll
  for (int i = 0; i < DEG; i++)
#pragma unroll
    for (int k = 0; k < ND; k++) { r[ND * i + k] = a.x.c[i].v[k]; r[DEG * ND + ND * i + k] = a.y.c[i].v[k]; }
  return r;
}
static PBC_DEV void f6_unpack(f6 &a, f6vec r) {
#pragma unroll
  for (int i = 0; i < DEG; i++)
#pragma unroll
    for (int k = 0; k < ND; k++) { a.x.c[i].v[k] = r[ND * i + k]; a.y.c[i].v[k] = r[DEG * ND + ND * i + k]; }
}
static PBC_DEV void f6_to_limbs(f6l &r, const f6 &a) {
#pragma unroll
  for (int i = 0; i < DEG; i++) { to_limbs<ND>(r.x[i], a.x.c[i]); to_limbs<ND>(r.y[i], a.y.c[i]); }
}
static PBC_DEV void f6_from_limbs(f6 &r, const f6l &a) {
#pragma unroll
  for (int i = 0; i < DEG; i++) { from_limbs<ND>(r.x.c[i], a.x[i]); from_limbs<ND>(r.y.c[i], a.y[i]); }
}
// coefficient S of A1 B1 + A2 B2 (plain polynomial products), plus the table terms for S < d.  DBL: how many of the
// operand pairs carry a doubled operand (limbs < 2^30: two units each)
template <int S, bool TABLE, int DBL>
static PBC_DEV void mul2_coeff(fl<ND> &out, const fl<ND> *A1, const fl<ND> *B1, const fl<ND> *A2, const fl<ND> *B2, const fl<ND> *H) {
  constexpr int lo = S - (DEG - 1) > 0 ? S - (DEG - 1) : 0, hi = S < DEG - 1 ? S : DEG - 1;
  constexpr int NDIR = hi - lo + 1, T = 2 * NDIR + (TABLE ? DEG - 1 : 0);
  fl<ND> x[T], y[T];
#pragma unroll
  for (int i = lo; i <= hi; i++) {
    x[2 * (i - lo)] = A1[i]; y[2 * (i - lo)] = B1[S - i];
    x[2 * (i - lo) + 1] = A2[i]; y[2 * (i - lo) + 1] = B2[S - i];
  }
  if constexpr (TABLE) {
#pragma unroll
    for (int t = 0; t < DEG - 1; t++) { x[2 * NDIR + t] = H[t]; y[2 * NDIR + t] = xpwr_limbs(t, S); }
  }
  sop_limbs<ND, T, DBL>(out, x, y);
}
// r = A1 B1 + A2 B2 in F_q^3 (limb forms in, P-class limb forms out).  DBL1: B1's coefficient 0 may be a sum of two.
template <bool DBL1>
static PBC_DEV void f3l_mul2(fl<ND> *r, const fl<ND> *A1, const fl<ND> *B1, const fl<ND> *A2, const fl<ND> *B2) {
  static_assert(DEG == 3, "written out for d = 3");
  fl<ND> H[2];
  mul2_coeff<3, false, 0>(H[0], A1, B1, A2, B2, H);
  mul2_coeff<4, false, 0>(H[1], A1, B1, A2, B2, H);
  mul2_coeff<0, true, DBL1 ? 1 : 0>(r[0], A1, B1, A2, B2, H);
  mul2_coeff<1, true, DBL1 ? 1 : 0>(r[1], A1, B1, A2, B2, H);
  mul2_coeff<2, true, DBL1 ? 1 : 0>(r[2], A1, B1, A2, B2, H);
}
static PBC_DEV void f3l_mul_v(fl<ND> *r, const fl<ND> *a) {
  fl<ND> V;
#pragma unroll
  for (int l = 0; l < Limbs29<ND>::L; l++) V.l[l] = c_d.nqr29[l];
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    const fl<ND> x[1] = {a[i]}, y[1] = {V};
    sop_limbs<ND, 1>(r[i], x, y);
  }
}
// r = a b.  BX0_DBL: b.x[0] is a sum of two normalised values (the "+ c" of a line value)
template <bool BX0_DBL>
static PBC_DEV void f6l_mul(f6l &r, const f6l &a, const f6l &b) {
  fl<ND> vay[DEG];
  f3l_mul_v(vay, a.y);
  f3l_mul2<BX0_DBL>(r.x, a.x, b.x, vay, b.y);
  // (ay bx + ax by: the possibly doubled b.x[0] must be the B1 operand)
  f3l_mul2<BX0_DBL>(r.y, a.y, b.x, a.x, b.y);
}
// r = a^2 = (ax^2 + (v ay) ay) + (2 ax) ay s: the two squares as products of a with itself (the cross terms are not
// folded: 4 more limb products per coefficient pair, no doubled operands to account for)
static PBC_DEV void f6l_sqr(f6l &r, const f6l &a) {
  fl<ND> vay[DEG], ax2[DEG];
  f3l_mul_v(vay, a.y);
  f3l_mul2<false>(r.x, a.x, a.x, vay, a.y);
#pragma unroll
  for (int i = 0; i < DEG; i++) limbs_dbl<ND>(ax2[i], a.x[i]);
  {                                              // (2 ax) ay: one polynomial product with a doubled operand
    fl<ND> H[2];
    static_assert(DEG == 3, "written out for d = 3");
    { const fl<ND> x[2] = {ax2[1], ax2[2]}, y[2] = {a.y[2], a.y[1]}; sop_limbs<ND, 2, 2>(H[0], x, y); }
    { const fl<ND> x[1] = {ax2[2]}, y[1] = {a.y[2]}; sop_limbs<ND, 1, 1>(H[1], x, y); }
    { const fl<ND> x[3] = {ax2[0], H[0], H[1]}, y[3] = {a.y[0], xpwr_limbs(0, 0), xpwr_limbs(1, 0)}; sop_limbs<ND, 3, 1>(r.y[0], x, y); }
    { const fl<ND> x[4] = {ax2[0], ax2[1], H[0], H[1]}, y[4] = {a.y[1], a.y[0], xpwr_limbs(0, 1), xpwr_limbs(1, 1)}; sop_limbs<ND, 4, 2>(r.y[1], x, y); }
    { const fl<ND> x[5] = {ax2[0], ax2[1], ax2[2], H[0], H[1]}, y[5] = {a.y[2], a.y[1], a.y[0], xpwr_limbs(0, 2), xpwr_limbs(1, 2)}; sop_limbs<ND, 5, 3>(r.y[2], x, y); }
  }
}
static __device__ __noinline__ f6vec f6_sqr_fused_fn(f6vec vv) {
  f6 v;
  f6l a, r;
  f6_unpack(v, vv);
  f6_to_limbs(a, v);
  f6l_sqr(r, a);
  f6_from_limbs(v, r);
  return f6_pack(v);
}
// ---- the point arithmetic on E(F_q) in limb form (kLimbPoint: d = 3 on the 5-word fields) ------------------------------
// X, Y, Z of the running point and the fixed P sit in LDS as 6-limb elements in a redundant representation (as in
// pairing_al.cuh: R = 2^174 against q < 2^160 leaves 14 bits of slack): additions are limb-wise without carries,
// a - b is a + K - b with K = c q in borrowed form (limb_i(c q) + D 2^29 - D dominates limbs up to D (2^29 - 1)), and a
// parallel carry pass renormalises where limbs would outgrow 32 bits or a column its 9.6 product units.  The bounds
// (u = limb size in units of 2^29, B = value in units of q) are noted per line and asserted in the host mirror, which
// also re-computes every column sum in 128 bits.  Stored: X, Y almost normalised with B <= 18, Z with u <= 2, B <= 3.
typedef fl<ND> el;
static constexpr int FLW = Limbs29<ND>::L;
enum { K2 = 0, K4 = 1, K16 = 2, K32 = 3 };                    // (c, D) = (2, 1), (4, 2), (16, 2), (32, 2)
static constexpr int KSUB_OFF = 64;                           // DConst::xpwr29 holds them behind the d = 3 table
static_assert(!kLimbPoint || ((DEG - 1) * DEG * FLW <= KSUB_OFF && KSUB_OFF + 5 * FLW <= 120), "room behind the x-power table");
// host mirror only: worst-case bound tracker, as in pairing_al.cuh -- every element carries the (u, B) its producing
// operations imply, every operation asserts its precondition on those: one run proves the bounds for all inputs
#ifdef PBC_HOSTSIM
static constexpr double U_STRICT = 1.0 - 1.0 / 536870912.0, U_ALMOST = 1.0 + 7.0 / 536870912.0;
static constexpr double KC[4] = {2, 4, 16, 32}, KD[4] = {1, 2, 2, 2};
static constexpr double SLACK = 16384.0;                      // R / q > 2^14
// classes of the five stored elements (X, Y, Z, Px, Py): what ll_put may store and what ll_get assumes -- the product
// kernel swaps the point state of its terms in and out of LDS behind the tracker's back
static constexpr double CLS_U[5] = {U_ALMOST, U_ALMOST, 2.0, U_STRICT, U_STRICT}, CLS_B[5] = {18, 18, 3.001, 1, 1};
static void hs_fail(const char *what, double v) { fprintf(stderr, "hostsim: limb-form type d point arithmetic: %s (%g)\n", what, v); abort(); }
static void hs_limbs(const el &a) {
  for (int i = 0; i < FLW; i++)
    if ((double) a.l[i] > a.hs_u * 536870912.0) hs_fail("limb above its tracked bound", a.hs_u);
}
static void hs_set(el &r, double u, double B) { r.hs_u = u; r.hs_B = B; hs_limbs(r); }
static void hs_dom(const el &b, int k) {
  hs_limbs(b);
  if (b.hs_u > KD[k] * U_STRICT + 1e-12) hs_fail("subtrahend limbs not dominated", b.hs_u);
  // top limb: K's is at least c q / 2^145 - 1 - D, b's at most B q / 2^145, and q >= 2^152 (init checks it: limb_ok)
  if ((KC[k] - b.hs_B) * 128.0 < KD[k] + 1) hs_fail("subtrahend value not dominated", b.hs_B);
}
static void hs_cols(double s) { if (s > (64.0 - FLW) / FLW - 0.01) hs_fail("column capacity", s); }
#define DL_HS(...) __VA_ARGS__
#else
#define DL_HS(...)
#endif
static PBC_DEV el l_const(int idx) {                          // idx 0..3: K constants, 4: the curve coefficient a
  el r;
#pragma unroll
  for (int l = 0; l < FLW; l++) r.l[l] = c_d.xpwr29[KSUB_OFF + idx * FLW + l];
  DL_HS(if (idx == 4) hs_set(r, U_STRICT, 1.0);)
  return r;
}
static PBC_DEV el ll_get(int w) {
  el r;
#pragma unroll
  for (int k = 0; k < FLW; k++) r.l[k] = g_lds_d<ND, DEG>[(w + k) * D_LANES + threadIdx.x];
  DL_HS(hs_set(r, CLS_U[(w - DL_X) / PW], CLS_B[(w - DL_X) / PW]);)
  return r;
}
static PBC_DEV void ll_put(int w, const el &a) {
#pragma unroll
  for (int k = 0; k < FLW; k++) g_lds_d<ND, DEG>[(w + k) * D_LANES + threadIdx.x] = a.l[k];
  DL_HS(hs_limbs(a); if (a.hs_u > CLS_U[(w - DL_X) / PW] || a.hs_B > CLS_B[(w - DL_X) / PW]) hs_fail("stored element outside its class", a.hs_B);)
}
static PBC_DEV void l_add(el &r, const el &a, const el &b) {
#pragma unroll
  for (int i = 0; i < FLW; i++) r.l[i] = a.l[i] + b.l[i];
  DL_HS(if (a.hs_u + b.hs_u >= 8) hs_fail("sum overflows 32 bits", a.hs_u + b.hs_u); hs_set(r, a.hs_u + b.hs_u, a.hs_B + b.hs_B);)
}
template <int S>
static PBC_DEV void l_shl(el &r, const el &a) {
#pragma unroll
  for (int i = 0; i < FLW; i++) r.l[i] = a.l[i] << S;
  DL_HS(if (a.hs_u * (1 << S) >= 8) hs_fail("shift overflows 32 bits", a.hs_u); hs_set(r, a.hs_u * (1 << S), a.hs_B * (1 << S));)
}
static PBC_DEV void l_subk(el &r, const el &a, const el &b, int k) {      // a + K_k - b
  const el K = l_const(k);
  DL_HS(hs_dom(b, k); const double u = a.hs_u + KD[k] + 1, B = a.hs_B + KC[k]; if (u >= 8) hs_fail("difference overflows 32 bits", u);)
#pragma unroll
  for (int i = 0; i < FLW; i++) r.l[i] = a.l[i] - b.l[i] + K.l[i];
  DL_HS(hs_set(r, u, B);)
}
static PBC_DEV void l_negk(el &r, const el &b, int k) {
  const el K = l_const(k);
  DL_HS(hs_dom(b, k);)
#pragma unroll
  for (int i = 0; i < FLW; i++) r.l[i] = K.l[i] - b.l[i];
  DL_HS(hs_set(r, KD[k] + 1, KC[k]);)
}
static PBC_DEV void l_norm(el &r, const el &a) {              // parallel carry pass: limbs <= 2^29 + 6
  uint32_t c = 0;
  DL_HS(hs_limbs(a); const double B = a.hs_B; if (a.hs_u >= 8) hs_fail("normalising limbs above 32 bits", a.hs_u);)
#pragma unroll
  for (int i = 0; i < FLW; i++) {
    const uint32_t t = a.l[i];
    r.l[i] = (i < FLW - 1 ? (t & Limbs29<ND>::MASK) : t) + c;
    c = t >> 29;
  }
  DL_HS(hs_set(r, U_ALMOST, B);)
}
// UNITS: the product of the operands' limb sizes (a column holds 9)
template <int UNITS>
static PBC_DEV void l_mul(el &r, const el &a, const el &b) {
  const el x[1] = {a}, y[1] = {b};
  DL_HS(hs_limbs(a); hs_limbs(b); hs_cols(a.hs_u * b.hs_u); if (a.hs_u * b.hs_u > UNITS + 0.001) hs_fail("more product units than declared", a.hs_u * b.hs_u);
        const double B = 1 + a.hs_B * b.hs_B / SLACK;)
  sop_limbs<ND, 1, UNITS - 1>(r, x, y);
  DL_HS(hs_set(r, U_STRICT, B);)
}
static PBC_DEV void l_sqr(el &r, const el &a) {               // a (almost) normalised
  DL_HS(const el x[1] = {a}; hs_limbs(a); if (a.hs_u > U_ALMOST) hs_fail("squaring an unnormalised element", a.hs_u); hs_sop_check<ND>(x, x, 1);
        const double B = 1 + a.hs_B * a.hs_B / SLACK;)
  sqr_limbs<ND>(r.l, a.l);
  DL_HS(hs_set(r, U_STRICT, B);)
}
// a0 b0 + a1 b1, one reduction
static PBC_DEV void l_sop2(el &r, const el &a0, const el &b0, const el &a1, const el &b1) {
  const el x[2] = {a0, a1}, y[2] = {b0, b1};
  DL_HS(hs_limbs(a0); hs_limbs(b0); hs_limbs(a1); hs_limbs(b1); hs_cols(a0.hs_u * b0.hs_u + a1.hs_u * b1.hs_u);
        if (a0.hs_u * b0.hs_u + a1.hs_u * b1.hs_u > 2.01) hs_fail("two-term sum of unnormalised operands", a0.hs_u);
        const double B = 1 + (a0.hs_B * b0.hs_B + a1.hs_B * b1.hs_B) / SLACK;)
  sop_limbs<ND, 2, 0>(r, x, y);
  DL_HS(hs_set(r, U_STRICT, B);)
}
// any class -> fully reduced words (one product by R mod q brings the value below 2q)
static PBC_DEV void l_to_fq(fq &r, const el &a) {
  el one, t;
  to_limbs<ND>(one, dk(fpk<ND>().one));
  DL_HS(hs_set(one, U_STRICT, 1.0);)
  l_mul<8>(t, a, one);
  from_limbs<ND>(r, t);
}
// tangent at V and V <- 2V (d_dbl_core): a' = -M Z^2 (u 2), b' = (2YZ) Z^2 (P-class), c' = M X - 2Y^2 (normalised)
static PBC_DEV void d_dbl_core_l(el &la, el &lb, el &lc) {
  const el X = ll_get(DL_X), Y = ll_get(DL_Y), Z = ll_get(DL_Z);
  el ZZ, XX, YY, M, t0, t1, S1, Z3, W, Xn, Yn;
  l_mul<4>(ZZ, Z, Z);
  l_sqr(XX, X);
  l_sqr(YY, Y);
  l_sqr(t0, ZZ);
  l_mul<1>(t0, t0, l_const(4));        // a Z^4
  l_shl<1>(M, XX);
  l_add(M, M, XX);
  l_add(M, M, t0);                     // u 4, B 5
  l_norm(M, M);
  l_mul<1>(la, M, ZZ);
  l_negk(la, la, K2);                  // u 2, B 2
  l_mul<2>(Z3, Y, Z);
  l_shl<1>(Z3, Z3);                    // 2YZ: u 2, B 3
  l_mul<2>(lb, Z3, ZZ);
  l_mul<1>(lc, M, X);
  l_shl<1>(t1, YY);                    // u 2, B 3
  l_subk(lc, lc, t1, K4);              // u 4, B 5.5
  l_norm(lc, lc);
  l_mul<1>(S1, X, YY);                 // X Y^2
  l_shl<3>(t1, S1);                    // 8 X Y^2: u < 8, B 12
  l_norm(t1, t1);
  l_sqr(t0, M);
  l_subk(t0, t0, t1, K16);             // X3 = M^2 - 2S, S = 4 X Y^2: u 4, B 17.5
  l_norm(Xn, t0);
  l_shl<2>(t1, S1);                    // S: u 4, B 6
  l_subk(W, t1, Xn, K32);              // S - X3: u 7, B 38
  l_mul<7>(t0, M, W);
  l_sqr(t1, YY);
  l_shl<3>(t1, t1);                    // 8 Y^4: u < 8, B 12
  l_norm(t1, t1);
  l_subk(t0, t0, t1, K16);             // Y3 = M (S - X3) - 8Y^4: u 4, B 17.5
  l_norm(Yn, t0);
  ll_put(DL_X, Xn);
  ll_put(DL_Y, Yn);
  ll_put(DL_Z, Z3);
}
// chord through V and +-P, V <- V +- P (d_add_core): a' = -R = Y - Py Z^3, b' = Z3, c' = R Px - Z3 Py
static PBC_DEV void d_add_core_l(el &la, el &lb, el &lc, bool neg) {
  const el X = ll_get(DL_X), Y = ll_get(DL_Y), Z = ll_get(DL_Z), Px = ll_get(DL_PX);
  el Py = ll_get(DL_PY);
  if (neg) {                           // wave-uniform
    l_negk(Py, Py, K2);
    l_norm(Py, Py);                    // B 2
  }
  el ZZ, H, Rn, HH, HHH, t0, t1, Z3, W, Xn, nY;
  l_mul<4>(ZZ, Z, Z);
  l_mul<1>(H, Px, ZZ);
  l_subk(H, H, X, K32);                // u 4, B 33.5
  l_norm(H, H);
  l_mul<2>(t0, Z, ZZ);
  l_mul<1>(t0, Py, t0);
  l_subk(Rn, Y, t0, K2);               // -R: u 4, B 19.5
  l_norm(Rn, Rn);
  l_mul<2>(Z3, Z, H);
  la = Rn;
  lb = Z3;
  l_sop2(t0, Rn, Px, Z3, Py);          // c' = -(Rn' Px + Z3 Py) with Rn' = -R: one lazy sum, then the negation
  // (R Px - Z3 Py with R = -Rn:  -(Rn Px) - Z3 Py)
  l_negk(lc, t0, K2);                  // u 2, B 2
  l_norm(lc, lc);
  l_sqr(HH, H);
  l_mul<1>(HHH, HH, H);
  l_mul<1>(t0, X, HH);                 // X1 H^2
  l_sqr(t1, Rn);
  l_subk(t1, t1, HHH, K2);             // u 3, B 3.5
  l_shl<1>(W, t0);                     // u 2, B 3
  l_subk(t1, t1, W, K4);               // X3 = R^2 - H^3 - 2 X1 H^2: u 6, B 7.5
  l_norm(Xn, t1);
  l_subk(W, Xn, t0, K2);               // X3 - X1 H^2: u 3, B 9.5
  l_norm(W, W);
  l_negk(nY, Y, K32);                  // -Y1: u 3, B 32
  l_norm(nY, nY);
  l_sop2(t0, Rn, W, nY, HHH);          // Y3 = R (X1 H^2 - X3) - Y1 H^3 = Rn (X3 - X1 H^2) + (-Y1) H^3
  ll_put(DL_X, Xn);
  ll_put(DL_Y, t0);
  ll_put(DL_Z, Z3);
}

// v * l(Q) for the line a' x + b' y + c' (d_miller_evalfn): l = (a' Qx + c') + (b' Qy) s, formed in limb form
// (La: limbs up to 2^30, Lb: P-class, Lc: normalised)
static PBC_DEV f6vec d_line_mul_l(f6vec vv, const fl<ND> &La, const fl<ND> &Lb, const fl<ND> &Lc) {
  DL_HS(if (kLimbPoint && (La.hs_u > 2.0 || Lb.hs_u > U_STRICT || Lc.hs_u > U_ALMOST || La.hs_B > 64 || Lb.hs_B > 64 || Lc.hs_B > 64))
          hs_fail("line coefficient outside its class", La.hs_u);)
  f6 v;
  f6l a, l, r;
  f6_unpack(v, vv);
  f6_to_limbs(a, v);
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    fl<ND> q;
    to_limbs<ND>(q, dl_get(DL_QX + ND * i));
    { const fl<ND> x[1] = {q}, y[1] = {La}; sop_limbs<ND, 1, 1>(l.x[i], x, y); }
    to_limbs<ND>(q, dl_get(DL_QY + ND * i));
    { const fl<ND> x[1] = {q}, y[1] = {Lb}; sop_limbs<ND, 1>(l.y[i], x, y); }
  }
#pragma unroll
  for (int k = 0; k < Limbs29<ND>::L; k++) l.x[0].l[k] += Lc.l[k];     // limbs < 2^30: the doubled operand of f6l_mul<true>
  if constexpr (!kLineDbl) {           // ... or one parallel carry pass where the column has no room for it
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < Limbs29<ND>::L; k++) {
      const uint32_t t = l.x[0].l[k];
      l.x[0].l[k] = (k < Limbs29<ND>::L - 1 ? (t & Limbs29<ND>::MASK) : t) + c;
      c = t >> 29;
    }
  }
  f6l_mul<kLineDbl>(r, a, l);
  f6_from_limbs(v, r);
  return f6_pack(v);
}
static PBC_DEV f6vec d_line_mul(f6vec vv, const fq &la, const fq &lb, const fq &lc) {
  fl<ND> La, Lb, Lc;
  d_line_to_limbs(La, Lb, Lc, la, lb, lc);
  return d_line_mul_l(vv, La, Lb, Lc);
}
// line coefficients of the word-form step routines as limbs
static PBC_DEV void d_line_to_limbs(el &La, el &Lb, el &Lc, const fq &la, const fq &lb, const fq &lc) {
  to_limbs<ND>(La, la);
  to_limbs<ND>(Lb, lb);
  to_limbs<ND>(Lc, lc);
  DL_HS(if (kLimbPoint) { hs_set(La, U_STRICT, 1.0); hs_set(Lb, U_STRICT, 1.0); hs_set(Lc, U_STRICT, 1.0); })
}
// (one copy of the fused product behind either form of the step routine: limb_ok is wave-uniform)
static __device__ __noinline__ f6vec d_dbl_line_mul_fn(f6vec v) {
  el La, Lb, Lc;
  bool limb = false;
  if constexpr (kLimbPoint) limb = c_d.limb_ok != 0;
  if (limb) {
    if constexpr (kLimbPoint) d_dbl_core_l(La, Lb, Lc);
  } else {
    fq la, lb, lc;
    d_dbl_core(la, lb, lc);
    d_line_to_limbs(La, Lb, Lc, la, lb, lc);
  }
  return d_line_mul_l(v, La, Lb, Lc);
}
static __device__ __noinline__ f6vec d_add_line_mul_fn(f6vec v, int neg) {
  el La, Lb, Lc;
  bool limb = false;
  if constexpr (kLimbPoint) limb = c_d.limb_ok != 0;
  if (limb) {
    if constexpr (kLimbPoint) d_add_core_l(La, Lb, Lc, neg != 0);
  } else {
    fq la, lb, lc;
    d_add_core(la, lb, lc, neg != 0);
    d_line_to_limbs(La, Lb, Lc, la, lb, lc);
  }
  return d_line_mul_l(v, La, Lb, Lc);
}
// digit of the Miller loop at position m: +1, -1 or 0 (wave-uniform; hostbn.h naf_of_half)
static PBC_DEV int d_digit(int m) {
  return (int) ((c_d.r[m >> 5] >> (m & 31)) & 1) - (int) ((c_d.rm[m >> 5] >> (m & 31)) & 1);
}

static PBC_DEV void f3_load_be(f3 &r, const uint8_t *src) { for (int i = 0; i < DEG; i++) fp_load_be<ND>(r.c[i], src + fpk<ND>().fbytes * i); }
static PBC_DEV void f3_store_be(uint8_t *dst, const f3 &a) { for (int i = 0; i < DEG; i++) fp_store_be<ND>(dst + fpk<ND>().fbytes * i, a.c[i]); }

// Per-lane set-up of one (P, Q) term: G1 bytes x||y (2 x fbytes), G2 bytes x||y over F_q^d (2 x d fbytes) -> the LDS
// Miller state (V = P, the fixed P, the untwisted Q).  Returns false when an input deserialises to O
// (curve_from_bytes, ecc/curve.c:609-623).
static PBC_DEV bool d_setup_lane(const uint8_t *g1, const uint8_t *g2) {
  const int NB = (int) fpk<ND>().fbytes;
  fq Px, Py, one;
  f3 Qx, Qy;
  fp_set<ND>(one, fpk<ND>().one);
  fp_load_be<ND>(Px, g1);
  fp_load_be<ND>(Py, g1 + NB);
  f3_load_be(Qx, g2);
  f3_load_be(Qy, g2 + DEG * NB);
  bool valid;
  {
    // curve_is_valid_point (curve.c:57-77): E: y^2 = x^3 + a x + b; twist over F_q^3
    fq t0, t1;
    fp_sqr<ND>(t0, Px);
    fp_add<ND>(t0, t0, dk(c_d.A));
    fp_mul<ND>(t0, t0, Px);
    fp_add<ND>(t0, t0, dk(c_d.B));
    fp_sqr<ND>(t1, Py);
    valid = fp_eq<ND>(t0, t1);
    f3 u0, u1;
    f3_sqr(u0, Qx);
    fp_add<ND>(u0.c[0], u0.c[0], dk(c_d.ta));
    f3_mul(u0, u0, Qx);
    fp_add<ND>(u0.c[0], u0.c[0], dk(c_d.tb));
    f3_sqr(u1, Qy);
    valid &= f3_eq(u0, u1);
  }
  // twist map (x, y) -> (v^-1 x, v^-2 y sqrt(v))  (cc_pairing, d_param.c:580-582)
  f3_mul_fq(Qx, Qx, dk(c_d.nqrinv));
  f3_mul_fq(Qy, Qy, dk(c_d.nqrinv2));
#pragma unroll
  for (int i = 0; i < DEG; i++) { dl_put(DL_QX + ND * i, Qx.c[i]); dl_put(DL_QY + ND * i, Qy.c[i]); }
  bool limb = false;
  if constexpr (kLimbPoint) limb = c_d.limb_ok != 0;
  if (limb) {
    if constexpr (kLimbPoint) {
      el x, y, o;
      to_limbs<ND>(x, Px); to_limbs<ND>(y, Py); to_limbs<ND>(o, one);
      DL_HS(hs_set(x, U_STRICT, 1.0); hs_set(y, U_STRICT, 1.0); hs_set(o, U_STRICT, 1.0);)
      ll_put(DL_X, x); ll_put(DL_Y, y); ll_put(DL_Z, o);
      ll_put(DL_PX, x); ll_put(DL_PY, y);
    }
  } else {
    dl_put(DL_X, Px); dl_put(DL_Y, Py); dl_put(DL_Z, one);
    dl_put(DL_PX, Px); dl_put(DL_PY, Py);
  }
  return valid;
}
// Miller function f_{r,P}(psi(Q)) of one term
static PBC_DEV bool d_miller_lane(f6 &v, const uint8_t *g1, const uint8_t *g2) {
  fq one;
  fp_set<ND>(one, fpk<ND>().one);
  const bool valid = d_setup_lane(g1, g2);
  f3_set_fq(v.x, one);
  f3_sub(v.y, v.x, v.x);
  // cc_miller_no_denom_affine (d_param.c:321-422): tangent; [double; line+add]; square
  if constexpr (kFusedF6) {
    f6vec vv = f6_pack(v);
    for (int m = c_d.rbits - 2;; m--) {
      d_fair_tick();
      vv = d_dbl_line_mul_fn(vv);
      if (m <= 0) break;
      const int dig = d_digit(m);
      if (dig) vv = d_add_line_mul_fn(vv, dig < 0);
      vv = f6_sqr_fused_fn(vv);
    }
    f6_unpack(v, vv);
    return valid;
  }
  for (int m = c_d.rbits - 2;; m--) {
    d_fair_tick();
    f6 e0;
    d_unpack(e0, d_dbl_line_fn());
    f6_mul(v, v, e0);
    if (m <= 0) break;
    const int dig = d_digit(m);
    if (dig) {
      d_unpack(e0, d_add_line_fn(dig < 0));
      f6_mul(v, v, e0);
    }
    f6_sqr(v, v);
  }
  return valid;
}

// cc_tatepower (d_param.c:505-564) with one inversion; see the header of this file.
static PBC_DEV void d_final_exp(f6 &out, const f6 &m) {
  const fq v = dk(c_d.nqr);
  // u = conj(m)^2 = (a^2 + v b^2) - 2ab sqrt(v),  N = a^2 - v b^2
  f3 aa, bb, ab, N;
  f6 u, uq, w;
  f3_sqr(aa, m.x);
  f3_sqr(bb, m.y);
  f3_mul_fq(bb, bb, v);
  f3_mul(ab, m.x, m.y);
  f3_add(u.x, aa, bb);
  f3_sub(N, aa, bb);
  f3_dbl(u.y, ab);
  f3_neg(u.y, u.y);
  // raise numerator and denominator to q + 1:  (x0 + x1 sqrt(v))^q = x0^q - x1^q sqrt(v)
  f3_frob(uq.x, u.x);
  f3_frob(uq.y, u.y);
  f3_neg(uq.y, uq.y);
  f6_mul(w, uq, u);                    // A + B sqrt(v)
  f3 D, t, invD, invB;
  f3_frob(D, N);
  f3_mul(D, D, N);
  // (B = 0 -- the value after the easy part is +-1, e.g. for the images of element_from_hash on GT, which lie in a product
  // of proper subfields -- must not poison 1/D: invert D * 1 instead.  The sqrt(v) part below is then
  // (P V_k - 2 V_{k-1}) D / (4 v) = 0 exactly, since P = +-2 gives V_n = 2 (+-1)^n.)
  f3 Bn = w.y;
  {
    f3 zero3, one3;
    fq o; fp_set<ND>(o, fpk<ND>().one);
    f3_set_fq(one3, o);
    f3_sub(zero3, one3, one3);
    const bool b0 = f3_eq(w.y, zero3);
#pragma unroll
    for (int i = 0; i < DEG; i++) fp_cmov<ND>(Bn.c[i], one3.c[i], b0);
  }
  f3_mul(t, D, Bn);
  f3_inv(t, t);                        // 1/(D B): the only inversion
  f3_mul(invD, t, Bn);
  f3_mul(invB, t, D);
  f3 h0, P, v0, v1, two;
  f3_mul(h0, w.x, invD);
  f3_dbl(P, h0);
  {
    fq o; fp_set<ND>(o, fpk<ND>().one); fp_dbl<ND>(o, o);
    f3_set_fq(two, o);
  }
  v0 = two;
  v1 = P;
  // lucas_even ladder (d_param.c:462-482): j == 0 takes the 0-branch
  for (int j = c_d.phikbits - 1; j >= 0; j--) {
    if ((j & 7) == 0) d_fair_tick();
    bool bit = j ? ((c_d.phik[j >> 5] >> (j & 31)) & 1) : false;
    f3 mm, s;
    f3_mul(mm, v0, v1);
    f3_sub(mm, mm, P);
    if (bit) { f3_sqr(s, v1); f3_sub(v1, s, two); v0 = mm; }
    else     { f3_sqr(s, v0); f3_sub(v0, s, two); v1 = mm; }
  }
  // cofactor odd: v1 = V_k, v0 = V_{k-1};  out = V_k/2 + (P V_k - 2 V_{k-1})/(P^2-4) * (B/D) sqrt(v)
  //                                            = V_k/2 + (P V_k - 2 V_{k-1}) D / (4 v B) sqrt(v)
  f3_mul(t, P, v1);
  f3_dbl(v0, v0);
  f3_sub(t, t, v0);
  f3_mul(t, t, D);
  f3_mul(t, t, invB);
  f3_mul_fq(t, t, dk(c_d.nqrinv));
  f3_halve(t, t);
  f3_halve(out.y, t);
  f3_halve(out.x, v1);
}

static PBC_DEV void d_store_gt(uint8_t *gt, f6 &out, bool valid) {
  if (!valid) {                        // GT identity
    fq one; fp_set<ND>(one, fpk<ND>().one);
    f3_set_fq(out.x, one);
    f3_sub(out.y, out.x, out.x);
  }
  f3_store_be(gt, out.x);
  f3_store_be(gt + DEG * fpk<ND>().fbytes, out.y);
}

// ---- preprocessed pairings: pairing_pp_init / pairing_pp_apply (include/pbc_pairing.h:54-89) ----
// d_pairing_pp_init (d_param.c:794-880) / g_pairing_pp_init (g_param.c:619-722) store the line
// coefficients of every Miller step for a fixed first argument; the apply routines (:908-966,
// :741-787) then need no arithmetic on E(F_q).  Here the table holds (a', b', c') of each step in the
// projective scaling of d_dbl_core / d_add_core, in loop order: [steps][3][ND] words, uniform data.
static PBC_DEV bool d_pp_init_lane(uint32_t *tab, const uint8_t *g1) {
  const int NB = (int) fpk<ND>().fbytes;
  fq Px, Py, one, t0, t1;
  fp_set<ND>(one, fpk<ND>().one);
  fp_load_be<ND>(Px, g1);
  fp_load_be<ND>(Py, g1 + NB);
  fp_sqr<ND>(t0, Px);
  fp_add<ND>(t0, t0, dk(c_d.A));
  fp_mul<ND>(t0, t0, Px);
  fp_add<ND>(t0, t0, dk(c_d.B));
  fp_sqr<ND>(t1, Py);
  bool valid = fp_eq<ND>(t0, t1);
  bool limb = false;
  if constexpr (kLimbPoint) limb = c_d.limb_ok != 0;
  if (limb) {
    if constexpr (kLimbPoint) {
      el x, y, o;
      to_limbs<ND>(x, Px); to_limbs<ND>(y, Py); to_limbs<ND>(o, one);
      DL_HS(hs_set(x, U_STRICT, 1.0); hs_set(y, U_STRICT, 1.0); hs_set(o, U_STRICT, 1.0);)
      ll_put(DL_X, x); ll_put(DL_Y, y); ll_put(DL_Z, o);
      ll_put(DL_PX, x); ll_put(DL_PY, y);
    }
  } else {
    dl_put(DL_X, Px); dl_put(DL_Y, Py); dl_put(DL_Z, one);
    dl_put(DL_PX, Px); dl_put(DL_PY, Py);
  }
  int slot = 0;
  for (int m = c_d.rbits - 2;; m--) {
    d_fair_tick();
    fq la, lb, lc;
    if (limb) {                        // the table stays in canonical word form
      if constexpr (kLimbPoint) {
        el a, b, c;
        d_dbl_core_l(a, b, c);
        l_to_fq(la, a); l_to_fq(lb, b); l_to_fq(lc, c);
      }
    } else {
      d_dbl_core(la, lb, lc);
    }
    for (int k = 0; k < ND; k++) { tab[(slot * 3 + 0) * ND + k] = la.v[k]; tab[(slot * 3 + 1) * ND + k] = lb.v[k]; tab[(slot * 3 + 2) * ND + k] = lc.v[k]; }
    slot++;
    if (m <= 0) break;
    if (d_digit(m)) {
      if (limb) {
        if constexpr (kLimbPoint) {
          el a, b, c;
          d_add_core_l(a, b, c, d_digit(m) < 0);
          l_to_fq(la, a); l_to_fq(lb, b); l_to_fq(lc, c);
        }
      } else {
        d_add_core(la, lb, lc, d_digit(m) < 0);
      }
      for (int k = 0; k < ND; k++) { tab[(slot * 3 + 0) * ND + k] = la.v[k]; tab[(slot * 3 + 1) * ND + k] = lb.v[k]; tab[(slot * 3 + 2) * ND + k] = lc.v[k]; }
      slot++;
    }
  }
  return valid;
}
static __device__ __noinline__ v32 d_pp_line_fn(v5 va, v5 vb, v5 vc) {
  fq a, b, c;
  from_vec<ND>(a, va);
  from_vec<ND>(b, vb);
  from_vec<ND>(c, vc);
  return d_evalfn_pack(a, b, c);
}
static PBC_DEV void d_pp_line(f6 &e0, const uint32_t *tab, int slot) {
  fq a, b, c;
#pragma unroll
  for (int k = 0; k < ND; k++) {
    a.v[k] = tab[(slot * 3 + 0) * ND + k];
    b.v[k] = tab[(slot * 3 + 1) * ND + k];
    c.v[k] = tab[(slot * 3 + 2) * ND + k];
  }
  d_unpack(e0, d_pp_line_fn(to_vec<ND>(a), to_vec<ND>(b), to_vec<ND>(c)));
}
// fused form (d159): the coefficients travel through the point slots of the LDS state, which pp_apply does not use
static __device__ __noinline__ f6vec d_pp_line_mul_fn(f6vec v) {
  return d_line_mul(v, dl_get(DL_X), dl_get(DL_Y), dl_get(DL_Z));
}
static PBC_DEV f6vec d_pp_line_mul(f6vec v, const uint32_t *tab, int slot) {
  fq a, b, c;
#pragma unroll
  for (int k = 0; k < ND; k++) {
    a.v[k] = tab[(slot * 3 + 0) * ND + k];
    b.v[k] = tab[(slot * 3 + 1) * ND + k];
    c.v[k] = tab[(slot * 3 + 2) * ND + k];
  }
  dl_put(DL_X, a); dl_put(DL_Y, b); dl_put(DL_Z, c);
  return d_pp_line_mul_fn(v);
}
// pairing_pp_apply for one lane
static PBC_DEV void d_pp_apply_lane(uint8_t *gt, const uint32_t *tab, bool p_valid, const uint8_t *g2) {
  const int NB = (int) fpk<ND>().fbytes;
  fq one;
  f3 Qx, Qy;
  f6 v, out;
  fp_set<ND>(one, fpk<ND>().one);
  f3_load_be(Qx, g2);
  f3_load_be(Qy, g2 + DEG * NB);
  bool valid = p_valid;
  {
    f3 u0, u1;
    f3_sqr(u0, Qx);
    fp_add<ND>(u0.c[0], u0.c[0], dk(c_d.ta));
    f3_mul(u0, u0, Qx);
    fp_add<ND>(u0.c[0], u0.c[0], dk(c_d.tb));
    f3_sqr(u1, Qy);
    valid &= f3_eq(u0, u1);
  }
  f3_mul_fq(Qx, Qx, dk(c_d.nqrinv));
  f3_mul_fq(Qy, Qy, dk(c_d.nqrinv2));
#pragma unroll
  for (int i = 0; i < DEG; i++) { dl_put(DL_QX + ND * i, Qx.c[i]); dl_put(DL_QY + ND * i, Qy.c[i]); }
  f3_set_fq(v.x, one);
  f3_sub(v.y, v.x, v.x);
  int slot = 0;
  if constexpr (kFusedF6) {
    f6vec vv = f6_pack(v);
    for (int m = c_d.rbits - 2;; m--) {
      d_fair_tick();
      vv = d_pp_line_mul(vv, tab, slot++);
      if (m <= 0) break;
      if (d_digit(m)) vv = d_pp_line_mul(vv, tab, slot++);
      vv = f6_sqr_fused_fn(vv);
    }
    f6_unpack(v, vv);
  } else {
    for (int m = c_d.rbits - 2;; m--) {
      d_fair_tick();
      f6 e0;
      d_pp_line(e0, tab, slot++);
      f6_mul(v, v, e0);
      if (m <= 0) break;
      if (d_digit(m)) {
        d_pp_line(e0, tab, slot++);
        f6_mul(v, v, e0);
      }
      f6_sqr(v, v);
    }
  }
  d_final_exp(out, v);
  d_store_gt(gt, out, valid);
}

// element_pairing (cc_pairing) / element_prod_pairing (cc_pairings_affine, d_param.c:710-736) for one lane.
// Products follow cc_millers_no_denom_affine (d_param.c:591-708): ONE accumulator is squared once per Miller iteration
// and takes the line values of all k terms, whose points advance in lockstep; then ONE cc_tatepower.  The LDS Miller
// state holds one term at a time; the others wait in a global workspace owned by the pairing object,
// word-major per 128-lane workgroup:  ws[((block k + term) DL_WORDS + word) 128 + lane]  (every access of a wave is
// 256 contiguous bytes).  Any identity input forces the product to 1.
static constexpr int DL_WORDS = 2 * DEG * ND + 5 * PW;
static PBC_DEV void d_ws_save(uint32_t *ws, int first, int count) {
#pragma unroll 5
  for (int w = first; w < first + count; w++) ws[w * D_LANES] = g_lds_d<ND, DEG>[w * D_LANES + threadIdx.x];
}
static PBC_DEV void d_ws_load(const uint32_t *ws, int first, int count) {
#pragma unroll 5
  for (int w = first; w < first + count; w++) g_lds_d<ND, DEG>[w * D_LANES + threadIdx.x] = ws[w * D_LANES];
}
static PBC_DEV void d_prod_pairing_lane(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, int k, uint32_t *ws) {
  f6 F, out;
  bool valid;
  if (k == 1) {
    valid = d_miller_lane(F, g1, g2);
  } else {
    const size_t L1 = 2 * fpk<ND>().fbytes, L2 = 2 * DEG * fpk<ND>().fbytes, REC = (size_t) DL_WORDS * D_LANES;
    valid = true;
    for (int j = 0; j < k; j++) {
      valid &= d_setup_lane(g1 + (size_t) j * L1, g2 + (size_t) j * L2);
      d_ws_save(ws + (size_t) j * REC, 0, DL_WORDS);
    }
    fq one;
    fp_set<ND>(one, fpk<ND>().one);
    f3_set_fq(F.x, one);
    f3_sub(F.y, F.x, F.x);
    f6vec FF = f6_pack(F);               // (kFusedF6: the accumulator in its register-vector form)
    for (int m = c_d.rbits - 2;; m--) {
      d_fair_tick();
      const int dig = m > 0 ? d_digit(m) : 0;
      for (int j = 0; j < k; j++) {
        uint32_t *w = ws + (size_t) j * REC;
        d_ws_load(w, 0, DL_WORDS);
        if constexpr (kFusedF6) {
          FF = d_dbl_line_mul_fn(FF);
          if (dig) FF = d_add_line_mul_fn(FF, dig < 0);
        } else {
          f6 e0;
          d_unpack(e0, d_dbl_line_fn());
          f6_mul(F, F, e0);
          if (dig) {
            d_unpack(e0, d_add_line_fn(dig < 0));
            f6_mul(F, F, e0);
          }
        }
        d_ws_save(w, DL_X, 3 * PW);      // only V = (X, Y, Z) changes
      }
      if (m <= 0) break;
      if constexpr (kFusedF6) FF = f6_sqr_fused_fn(FF);
      else f6_sqr(F, F);
    }
    if constexpr (kFusedF6) f6_unpack(F, FF);
  }
  d_final_exp(out, F);
  d_store_gt(gt, out, valid);
}

// ---- device-side derivation of the tower constants (host supplies canonical words) -------
// stage 1: everything that needs only F_q arithmetic
static PBC_DEV void init_stage1(DConst *out, const DRaw &raw, const DConst &base) {
  DConst C = base;
  fq r2, t, a, b, v, cf[DEG];
  fp_set<ND>(r2, fpk<ND>().r2);
  fp_set<ND>(t, raw.a); fp_mul<ND>(a, t, r2);
  fp_set<ND>(t, raw.b); fp_mul<ND>(b, t, r2);
  fp_set<ND>(t, raw.nqr); fp_mul<ND>(v, t, r2);
  for (int i = 0; i < DEG; i++) { fp_set<ND>(t, raw.coeff[i]); fp_mul<ND>(cf[i], t, r2); }
  // x^d = -(c0 + c1 x + ...);  x^(d+j) = x * x^(d+j-1) reduced (compute_x_powers, poly.c:1302-1333)
  fq xp[DEG - 1][DEG];
  for (int i = 0; i < DEG; i++) fp_neg<ND>(xp[0][i], cf[i]);
  for (int j = 1; j < DEG - 1; j++) {
    fp_mul<ND>(xp[j][0], xp[j - 1][DEG - 1], xp[0][0]);
    for (int i = 1; i < DEG; i++) {
      fp_mul<ND>(t, xp[j - 1][DEG - 1], xp[0][i]);
      fp_add<ND>(xp[j][i], xp[j - 1][i - 1], t);
    }
  }
  fq vi, vi2, v2, ta, tb;
  fp_inv<ND>(vi, v);
  fp_sqr<ND>(vi2, vi);
  fp_sqr<ND>(v2, v);
  fp_mul<ND>(ta, a, v2);
  fp_mul<ND>(v2, v2, v);
  fp_mul<ND>(tb, b, v2);
  for (int k = 0; k < ND; k++) {
    C.A[k] = a.v[k]; C.B[k] = b.v[k]; C.nqr[k] = v.v[k]; C.nqrinv[k] = vi.v[k]; C.nqrinv2[k] = vi2.v[k];
    C.ta[k] = ta.v[k]; C.tb[k] = tb.v[k];
  }
  for (int j = 0; j < DEG - 1; j++)
    for (int i = 0; i < DEG; i++) {
      fl<ND> t29;
      to_limbs<ND>(t29, xp[j][i]);
      for (int l = 0; l < Limbs29<ND>::L; l++) C.xpwr29[(j * DEG + i) * Limbs29<ND>::L + l] = t29.l[l];
    }
  {
    fl<ND> t29;
    to_limbs<ND>(t29, v);
    for (int l = 0; l < Limbs29<ND>::L; l++) C.nqr29[l] = t29.l[l];
  }
  if constexpr (kLimbPoint) {
    // subtraction constants of the limb-form point arithmetic: c q in 29-bit limbs, borrowed so that limb i
    // dominates limbs up to D (2^29 - 1):  limb_0 + D 2^29,  limb_i + D 2^29 - D,  limb_top - D;  then the curve's a
    const uint32_t cd[4][2] = {{2, 1}, {4, 2}, {16, 2}, {32, 2}};
    for (int t = 0; t < 4; t++) {
      uint64_t carry = 0;
      for (int i = 0; i < FLW; i++) {
        const uint64_t x = (uint64_t) cd[t][0] * fpk<ND>().p29[i] + carry;
        uint32_t k = (uint32_t) x & Limbs29<ND>::MASK;
        carry = x >> 29;
        k += (i < FLW - 1 ? cd[t][1] << 29 : 0) - (i > 0 ? cd[t][1] : 0);
        C.xpwr29[KSUB_OFF + t * FLW + i] = k;
      }
    }
    fl<ND> a29;
    to_limbs<ND>(a29, a);
    for (int l = 0; l < FLW; l++) C.xpwr29[KSUB_OFF + 4 * FLW + l] = a29.l[l];
  }
  *out = C;
}
// stage 2 (c_d now holds stage 1): x^q by square-and-multiply in F_q^d, then its powers
static PBC_DEV void init_stage2(DConst *out, const DRaw &raw) {
  DConst C = c_d;
  f3 acc, x, pw;
  fq one, zero;
  fp_set<ND>(one, fpk<ND>().one);
  for (int k = 0; k < ND; k++) zero.v[k] = 0;
  f3_set_fq(acc, one);
  f3_set_fq(x, zero);
  x.c[1] = one;
  for (int i = raw.qbits - 1; i >= 0; i--) {
    f3_sqr(acc, acc);
    if ((raw.q[i >> 5] >> (i & 31)) & 1) f3_mul(acc, acc, x);
  }
  pw = acc;
  for (int j = 0; j < DEG - 1; j++) {
    for (int i = 0; i < DEG; i++)
      for (int k = 0; k < ND; k++) C.xpowq[j][i][k] = pw.c[i].v[k];
    f3_mul(pw, pw, acc);
  }
  *out = C;
}
};  // struct TypeMNT

template <int ND> using TypeD = TypeMNT<ND, 3>;   // MNT, k = 6
typedef TypeMNT<5, 5> TypeG;                      // Freeman, k = 10 (g149.param: 149-bit q)

}  // namespace pbc
