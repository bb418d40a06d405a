// group_ops.cuh -- batched group operations next to the pairing path:
// scalar multiplication in G1 / G2 = E(F_q) and products / powers in GT, one element per lane.
//
// Reference: element_mul_zn / element_pow_zn (include/pbc_field.h:311, :374) ->
// field->mul_mpz / pow_mpz = generic_pow_mpz (arith/field.c:113-126, sliding window) over the
// group law curve_mul / curve_double (ecc/curve.c:102-207, affine, one inversion per step), and
// element_mul on GT (the mulg wrapper, ecc/pairing.c:135-283 -> fi_mul / fq_mul / polymod_mul).
// Group elements are unique, so any addition chain gives the same bytes.  Here: Jacobian
// double-and-always-add with a per-lane select (lanes hold different scalars; control flow stays
// wave-uniform), one safegcd inversion at the end.
#pragma once
#include "fp.cuh"
#include "pairing_a.cuh"
#include "pairing_d.cuh"
#include "pairing_f.cuh"
#include "pairing_e.cuh"

namespace pbc {

struct CurveK {                        // E: y^2 = x^3 + a x + b over F_q (Montgomery words)
  uint32_t a[34], b[34];
  int a_is_zero;
  // element_from_hash on G1: cofactor (curve_data cofac, ecc/curve.c:478) and square roots in F_q
  uint32_t cofac[34];                  // 0 bits: no cofactor multiplication (type f)
  int cofbits;
  int sqrt_mode;                       // 0: q = 3 mod 4, root = t^((q+1)/4);  1: Tonelli-Shanks
  uint32_t sqrt_e[34];                 // mode 0: (q+1)/4;  mode 1: (t-1)/2 with q - 1 = 2^s t, t odd
  int sqrt_bits;
  int ts_s;
  uint32_t ts_c[34];                   // mode 1: z^t for a non-residue z (Montgomery form; derived on the device)
};
static_assert(sizeof(CurveK) <= KOFF_TYPE - KOFF_CURVE, "constant block layout");
#define c_curve (pbc::kconst<pbc::CurveK, pbc::KOFF_CURVE>())

// bit i of a big-endian scalar of zlen bytes
PBC_DEV uint32_t zr_bit(const uint8_t *z, int zlen, int i) { return (z[zlen - 1 - (i >> 3)] >> (i & 7)) & 1; }

// ---- scalar multiplication on y^2 = x^3 + a x + b over a field given by a policy F ------------
// F supplies: el, bytes(), curve_a/curve_b/a_is_zero(), one/zero, add/sub/dbl/mul/sqr/inv, is0/eq,
// cmov, load/store.  Used for E(F_q) (G1; G2 of the symmetric types) and for the twists over
// F_q^d (types d, g: curve.c:885-901 coefficients a v^2, b v^3) and F_q^2 (type f: f_param.c:372-383).
template <int N> PBC_DEV void fq_from_hash_lane(fp<N> &x, const uint8_t *data, int hlen);   // defined below
template <int N> PBC_DEV void fp_sqrt_lane(fp<N> &y, bool &ok, const fp<N> &t);

template <int N>
struct FqOps {
  typedef fp<N> el;
  static constexpr int NW = N;         // words of F_q (selects the kernels' constant block, KArgs<NW>)
  static PBC_DEV int bytes() { return (int) fpk<N>().fbytes; }
  static PBC_DEV el curve_a() { el r; fp_set<N>(r, c_curve.a); return r; }
  static PBC_DEV el curve_b() { el r; fp_set<N>(r, c_curve.b); return r; }
  static PBC_DEV bool a_is_zero() { return c_curve.a_is_zero != 0; }
  static PBC_DEV el one() { el r; fp_set<N>(r, fpk<N>().one); return r; }
  static PBC_DEV el zero() { el r; for (int k = 0; k < N; k++) r.v[k] = 0; return r; }
  static PBC_DEV void add(el &r, const el &a, const el &b) { fp_add<N>(r, a, b); }
  static PBC_DEV void sub(el &r, const el &a, const el &b) { fp_sub<N>(r, a, b); }
  static PBC_DEV void dbl(el &r, const el &a) { fp_dbl<N>(r, a); }
  static PBC_DEV void mul(el &r, const el &a, const el &b) { fp_mul<N>(r, a, b); }
  static PBC_DEV void sqr(el &r, const el &a) { fp_sqr<N>(r, a); }
  static PBC_DEV void inv(el &r, const el &a) { fp_inv<N>(r, a); }
  static PBC_DEV bool is0(const el &a) { return fp_is0<N>(a); }
  static PBC_DEV bool eq(const el &a, const el &b) { return fp_eq<N>(a, b); }
  static PBC_DEV void cmov(el &r, const el &a, bool c) { fp_cmov<N>(r, a, c); }
  static PBC_DEV void load(el &r, const uint8_t *s) { fp_load_be<N>(r, s); }
  static PBC_DEV void store(uint8_t *d, const el &a) { fp_store_be<N>(d, a); }
  static constexpr int WORDS_EL = N;   // Montgomery words of an element (fixed-base tables)
  static PBC_DEV el from_words(const uint32_t *w) { el r; fp_set<N>(r, w); return r; }
  static PBC_DEV void to_words(uint32_t *w, const el &a) { for (int k = 0; k < N; k++) w[k] = a.v[k]; }
};
template <int N, int DEG>
struct FdOps {                         // F_q^d of types d / g: the twist E'(F_q^d)
  typedef TypeMNT<N, DEG> T;
  typedef typename T::f3 el;
  static constexpr int NW = N;
  static PBC_DEV int bytes() { return DEG * (int) fpk<N>().fbytes; }
  static PBC_DEV el curve_a() { el r; T::f3_set_fq(r, T::dk(c_d.ta)); return r; }
  static PBC_DEV el curve_b() { el r; T::f3_set_fq(r, T::dk(c_d.tb)); return r; }
  static PBC_DEV bool a_is_zero() { return false; }
  static PBC_DEV el one() { el r; fp<N> o; fp_set<N>(o, fpk<N>().one); T::f3_set_fq(r, o); return r; }
  static PBC_DEV el zero() { el r = one(); T::f3_sub(r, r, r); return r; }
  static PBC_DEV void add(el &r, const el &a, const el &b) { T::f3_add(r, a, b); }
  static PBC_DEV void sub(el &r, const el &a, const el &b) { T::f3_sub(r, a, b); }
  static PBC_DEV void dbl(el &r, const el &a) { T::f3_dbl(r, a); }
  static PBC_DEV void mul(el &r, const el &a, const el &b) { T::f3_mul(r, a, b); }
  static PBC_DEV void sqr(el &r, const el &a) { T::f3_sqr(r, a); }
  static PBC_DEV void inv(el &r, const el &a) { T::f3_inv(r, a); }
  static PBC_DEV bool is0(const el &a) { el z = zero(); return T::f3_eq(a, z); }
  static PBC_DEV bool eq(const el &a, const el &b) { return T::f3_eq(a, b); }
  static PBC_DEV void cmov(el &r, const el &a, bool c) { for (int i = 0; i < DEG; i++) fp_cmov<N>(r.c[i], a.c[i], c); }
  static PBC_DEV void load(el &r, const uint8_t *s) { T::f3_load_be(r, s); }
  static PBC_DEV void store(uint8_t *d, const el &a) { T::f3_store_be(d, a); }
  // extension-field helpers for element_from_hash / compressed points on the twist
  static constexpr int WORDS = DEG * N, WORDS_EL = DEG * N;
  static PBC_DEV el from_words(const uint32_t *w) { el r; for (int i = 0; i < DEG; i++) fp_set<N>(r.c[i], w + N * i); return r; }
  static PBC_DEV void to_words(uint32_t *w, const el &a) { for (int i = 0; i < DEG; i++) for (int k = 0; k < N; k++) w[N * i + k] = a.c[i].v[k]; }
  static PBC_DEV el nonresidue() { el r; T::f3_set_fq(r, T::dk(c_d.nqr)); return r; }   // v: F_q^k = F_q^d[sqrt(v)]
  // polymod_from_hash (arith/poly.c:341-348): every coefficient is the same hash value
  static PBC_DEV el hash_x(const uint8_t *data, int hlen) {
    el r;
    fq_from_hash_lane<N>(r.c[0], data, hlen);
    for (int i = 1; i < DEG; i++) r.c[i] = r.c[0];
    return r;
  }
  // element_sqrt as the x-only format sees it: polymod_sqrt (poly.c:634-700) is randomised, any root is "the" root
  static PBC_DEV void sqrt_ref(el &y, bool &ok, const el &t);
  // polymod_sgn (poly.c:1189-1199) with fp_sgn_odd (montfp.c:460-472): the sign of the first non-zero
  // coefficient, positive when its canonical residue is odd.  Returns true for "negative".
  static PBC_DEV bool is_negative(const el &a) {
    fp<N> o, c;
    for (int k = 0; k < N; k++) o.v[k] = (k == 0);
    bool decided = false, negative = false;
    for (int i = 0; i < DEG; i++) {
      fp_mul<N>(c, a.c[i], o);
      const bool nz = !fp_is0<N>(c);
      negative = (!decided & nz) ? ((c.v[0] & 1) == 0) : negative;
      decided |= nz;
    }
    return negative;
  }
};
template <int ND>
struct Fq2Ops {                        // F_q^2 of type f: the twist y^2 = x^3 + tb
  typedef TypeF<ND> T;
  typedef typename T::g2 el;
  static constexpr int NW = ND;
  static PBC_DEV int bytes() { return 2 * (int) fpk<ND>().fbytes; }
  static PBC_DEV el curve_a() { el r; T::g2_zero(r); return r; }
  static PBC_DEV el curve_b() { return T::fk2(c_f.tb); }
  static PBC_DEV bool a_is_zero() { return true; }
  static PBC_DEV el one() { el r; T::g2_zero(r); fp_set<ND>(r.x, fpk<ND>().one); return r; }
  static PBC_DEV el zero() { el r; T::g2_zero(r); return r; }
  static PBC_DEV void add(el &r, const el &a, const el &b) { T::g2_add(r, a, b); }
  static PBC_DEV void sub(el &r, const el &a, const el &b) { T::g2_sub(r, a, b); }
  static PBC_DEV void dbl(el &r, const el &a) { T::g2_dbl(r, a); }
  static PBC_DEV void mul(el &r, const el &a, const el &b) { T::g2_mul(r, a, b); }
  static PBC_DEV void sqr(el &r, const el &a) { T::g2_sqr(r, a); }
  static PBC_DEV void inv(el &r, const el &a) { T::g2_inv(r, a); }
  static PBC_DEV bool is0(const el &a) { return (int) fp_is0<ND>(a.x) & (int) fp_is0<ND>(a.y); }
  static PBC_DEV bool eq(const el &a, const el &b) { return T::g2_eq(a, b); }
  static PBC_DEV void cmov(el &r, const el &a, bool c) { fp_cmov<ND>(r.x, a.x, c); fp_cmov<ND>(r.y, a.y, c); }
  static PBC_DEV void load(el &r, const uint8_t *s) { T::g2_load_be(r, s); }
  static PBC_DEV void store(uint8_t *d, const el &a) { T::g2_store_be(d, a); }
  static constexpr int WORDS = 2 * ND, WORDS_EL = 2 * ND;
  static PBC_DEV el from_words(const uint32_t *w) { el r; fp_set<ND>(r.x, w); fp_set<ND>(r.y, w + ND); return r; }
  static PBC_DEV void to_words(uint32_t *w, const el &a) { for (int k = 0; k < ND; k++) { w[k] = a.x.v[k]; w[ND + k] = a.y.v[k]; } }
  // x^6 + alpha is irreducible, so alpha is no square in F_q^2, and neither is -alpha (-1 is a square there)
  static PBC_DEV el nonresidue() { return T::fk2(c_f.negalpha); }
  // fq_from_hash (arith/fieldquadratic.c:311-316): the two halves of the digest
  static PBC_DEV el hash_x(const uint8_t *data, int hlen) {
    el r;
    const int k = hlen / 2;
    fq_from_hash_lane<ND>(r.x, data, k);
    fq_from_hash_lane<ND>(r.y, data + k, hlen - k);
    return r;
  }
  // fq_sqrt (fieldquadratic.c:357-392): a + b sqrt(beta) with 2a^2 = x +- sqrt(x^2 - beta y^2), 2ab = y; the root
  // it returns is a function of the roots element_sqrt returns in F_q (defined for q = 3 mod 4)
  static PBC_DEV void sqrt_ref(el &r, bool &ok, const el &t) {
    fp<ND> e0, e1, e2, s0, half, beta;
    bool ok0, ok1, ok2;
    fp_set<ND>(beta, c_f.beta);
    fp_sqr<ND>(e0, t.x);
    fp_sqr<ND>(e1, t.y);
    fp_mul<ND>(e1, e1, beta);
    fp_sub<ND>(e0, e0, e1);
    fp_sqrt_lane<ND>(s0, ok0, e0);                   // sqrt(x^2 - beta y^2): exists iff t is a square
    fp_set<ND>(half, fpk<ND>().one);
    fp_halve<ND>(half, half);
    fp_add<ND>(e1, t.x, s0);
    fp_mul<ND>(e1, e1, half);
    fp_sqrt_lane<ND>(e2, ok1, e1);                   // is_sqr(e1) and its root in one go
    fp<ND> alt, a2;
    fp_sub<ND>(alt, e1, s0);
    fp_sqrt_lane<ND>(a2, ok2, alt);
    fp_cmov<ND>(e2, a2, !ok1);
    ok = ok0 & (ok1 | ok2);
    fp_dbl<ND>(e1, e2);
    fp_inv<ND>(e1, e1);
    fp_mul<ND>(r.y, t.y, e1);
    r.x = e2;
  }
  // fq_sign (fieldquadratic.c:159-165): sign of x, of y when x = 0
  static PBC_DEV bool is_negative(const el &a) {
    fp<ND> o, cx, cy;
    for (int k = 0; k < ND; k++) o.v[k] = (k == 0);
    fp_mul<ND>(cx, a.x, o);
    fp_mul<ND>(cy, a.y, o);
    const bool xz = fp_is0<ND>(cx), yz = fp_is0<ND>(cy);
    return xz ? (!yz & ((cy.v[0] & 1) == 0)) : ((cx.v[0] & 1) == 0);
  }
};

// Jacobian steps shared by the scalar multiplication and the cofactor ladder of element_from_hash.
// R <- 2R (dbl-2007-bl shape; Z = 0 stays 0, and a point with Y = 0 doubles to Z = 0, i.e. O)
template <class F>
PBC_DEV void ec_dbl_jac(typename F::el &X, typename F::el &Y, typename F::el &Z, const typename F::el &ca) {
  typedef typename F::el el;
  el XX, YY, ZZ, M, S, t0, t1, Z3;
  F::sqr(XX, X);
  F::sqr(YY, Y);
  F::sqr(ZZ, Z);
  F::dbl(M, XX);
  F::add(M, M, XX);
  if (!F::a_is_zero()) {
    F::sqr(t0, ZZ);
    F::mul(t0, t0, ca);
    F::add(M, M, t0);
  }
  F::mul(Z3, Y, Z);
  F::dbl(Z3, Z3);
  F::mul(S, X, YY);
  F::dbl(S, S);
  F::dbl(S, S);
  F::sqr(t0, YY);
  F::dbl(t0, t0);
  F::dbl(t0, t0);
  F::dbl(t0, t0);
  F::sqr(X, M);
  F::dbl(t1, S);
  F::sub(X, X, t1);
  F::sub(t1, S, X);
  F::mul(t1, M, t1);
  F::sub(Y, t1, t0);
  Z = Z3;
}
// R <- R + P when `take`, for the affine P = (x2, y2) with 2P = (DX, DY, DZ) precomputed.  Complete: R = O gives P,
// R = -P gives O (Z3 = Z H = 0), and R = P -- which the chord formulas cannot express (H = R = 0) -- takes the
// precomputed double.  The last case occurs for points whose order divides a prefix of the scalar minus one:
// curve_from_bytes (ecc/curve.c:609-623) accepts every point of the curve, not only the order-r subgroup, and the
// reference's curve_mul (curve.c:153-207) tests x1 == x2 on every addition.
template <class F>
PBC_DEV void ec_madd_jac(typename F::el &X, typename F::el &Y, typename F::el &Z, const typename F::el &x2,
                         const typename F::el &y2, const typename F::el &DX, const typename F::el &DY,
                         const typename F::el &DZ, bool take) {
  typedef typename F::el el;
  el ZZ, H, R, HH, HHH, t0, t1, X3, Y3, Z3;
  F::sqr(ZZ, Z);
  F::mul(H, x2, ZZ);
  F::sub(H, H, X);
  F::mul(t0, Z, ZZ);
  F::mul(R, y2, t0);
  F::sub(R, R, Y);
  F::mul(Z3, Z, H);
  F::sqr(HH, H);
  F::mul(HHH, HH, H);
  F::mul(t0, X, HH);
  F::sqr(X3, R);
  F::sub(X3, X3, HHH);
  F::sub(X3, X3, t0);
  F::sub(X3, X3, t0);
  F::sub(t0, t0, X3);
  F::mul(t0, R, t0);
  F::mul(t1, Y, HHH);
  F::sub(Y3, t0, t1);
  const bool inf = F::is0(Z);
  const bool same = !inf & F::is0(H) & F::is0(R);
  const bool take_p = take & inf, take_d = take & same, take_t = take & !inf & !same;
  F::cmov(X, X3, take_t);
  F::cmov(Y, Y3, take_t);
  F::cmov(Z, Z3, take_t);
  F::cmov(X, x2, take_p);
  F::cmov(Y, y2, take_p);
  F::cmov(Z, F::one(), take_p);
  F::cmov(X, DX, take_d);
  F::cmov(Y, DY, take_d);
  F::cmov(Z, DZ, take_d);
}

// out = [k] P for P = (x, y) bytes; off-curve P is O (curve_from_bytes); O serialises as zeros.
// Double-and-always-add with per-lane selects (element_mul_zn -> generic_pow_mpz over curve_mul,
// arith/field.c:113-126, ecc/curve.c:153-207).  Any scalar of zlen bytes and any point of the curve are
// handled (scalars >= r, points outside the order-r subgroup): the group law is complete here.
template <class F>
PBC_DEV void ec_mul_lane(uint8_t *out, const uint8_t *in, const uint8_t *z, int zlen) {
  typedef typename F::el el;
  const int NB = F::bytes();
  const el one = F::one(), ca = F::curve_a(), cb = F::curve_b();
  el x2, y2;
  F::load(x2, in);
  F::load(y2, in + NB);
  bool valid;
  {
    el t0, t1;
    F::sqr(t0, x2);
    F::add(t0, t0, ca);
    F::mul(t0, t0, x2);
    F::add(t0, t0, cb);
    F::sqr(t1, y2);
    valid = F::eq(t0, t1);
  }
  el DX = x2, DY = y2, DZ = one;       // 2P, for the step that meets R = P
  ec_dbl_jac<F>(DX, DY, DZ, ca);
  el X = one, Y = one, Z = F::zero();  // accumulator R, starts at O (Z = 0)
  for (int i = 8 * zlen - 1; i >= 0; i--) {
    ec_dbl_jac<F>(X, Y, Z, ca);
    ec_madd_jac<F>(X, Y, Z, x2, y2, DX, DY, DZ, zr_bit(z, zlen, i) != 0);
  }
  // to affine: x = X/Z^2, y = Y/Z^3
  el zi, zi2, ax, ay;
  bool is_inf = F::is0(Z) | !valid;
  F::inv(zi, Z);
  F::sqr(zi2, zi);
  F::mul(ax, X, zi2);
  F::mul(zi2, zi2, zi);
  F::mul(ay, Y, zi2);
  if (is_inf) { ax = F::zero(); ay = ax; }
  F::store(out, ax);
  F::store(out + NB, ay);
}
template <int N>
PBC_DEV void g_mul_lane(uint8_t *out, const uint8_t *in, const uint8_t *z, int zlen) {
  ec_mul_lane<FqOps<N>>(out, in, z, zlen);
}

// ---- the fast ladders: regular signed fixed windows and fixed-base tables ----------------------------------------------
// The routines above are COMPLETE (any point of the curve, any scalar) and pay for it: two group operations per scalar
// bit.  The library runs them only for lanes the routines below report.  Those use the plain Jacobian formulas -- an
// addition that meets R = +-T or R = O has H = 0, which leaves Z = 0 for good -- and report "Z = 0 at the end": the true
// result O, points of small order, exceptional scalars.  For points of the order-r subgroup and scalars below r that is
// a probability-2^-150 event, so the complete pass is a second, nearly empty launch.

// R <- R + (x2, y2), mixed, incomplete (ec_madd_jac without the case analysis)
template <class F>
PBC_DEV void ec_madd_inc(typename F::el &X, typename F::el &Y, typename F::el &Z, const typename F::el &x2,
                         const typename F::el &y2) {
  typedef typename F::el el;
  el ZZ, H, R, HH, HHH, t0, t1;
  F::sqr(ZZ, Z);
  F::mul(H, x2, ZZ);
  F::sub(H, H, X);
  F::mul(t0, Z, ZZ);
  F::mul(R, y2, t0);
  F::sub(R, R, Y);
  F::mul(Z, Z, H);
  F::sqr(HH, H);
  F::mul(HHH, HH, H);
  F::mul(t0, X, HH);
  F::sqr(X, R);
  F::sub(X, X, HHH);
  F::sub(X, X, t0);
  F::sub(X, X, t0);
  F::sub(t0, t0, X);
  F::mul(t0, R, t0);
  F::mul(t1, Y, HHH);
  F::sub(Y, t0, t1);
}
// the scalar as little-endian words in private memory (read once), one zero word above
constexpr int kScalarWords = 36;
PBC_DEV void scalar_load(uint32_t *kw, const uint8_t *z, int zlen) {
  for (int w = 0; w < kScalarWords; w++) kw[w] = 0;
  for (int i = 0; i < zlen; i++) kw[i >> 2] |= (uint32_t) z[zlen - 1 - i] << (8 * (i & 3));
}
PBC_DEV uint32_t scalar_bits(const uint32_t *kw, int pos, uint32_t mask) {
  const uint64_t pair = ((uint64_t) kw[(pos >> 5) + 1] << 32) | kw[pos >> 5];
  return (uint32_t) (pair >> (pos & 31)) & mask;
}
// out = [k] P by a regular signed fixed-window ladder, w = 4 (the algorithm of group_al.cuh, which states it in full, on a
// field policy): digits d_i = 2 ((k' >> (4 i + 1)) & 15) - 15 of k' = k | 1 with the bit above the scalar set, a per-lane
// table {P, 3P, ..., 15P} of affine points built on the curve where 2P is affine and normalised by one batched
// inversion, four doublings and one mixed addition per window, [k] P = [k + 1] P - P for even k.  Returns false (nothing
// written) when the lane needs the complete routine.
template <class F>
PBC_DEV bool ec_mul_win_lane(uint8_t *out, const uint8_t *in, const uint8_t *z, int zlen) {
  typedef typename F::el el;
  constexpr int WIN = 4, TE = 8;
  const int NB = F::bytes();
  const el one = F::one(), ca = F::curve_a(), cb = F::curve_b();
  el tab[TE][2], zs[TE], cs[TE];
  F::load(tab[0][0], in);
  F::load(tab[0][1], in + NB);
  bool valid;
  {
    el t0, t1;
    F::sqr(t0, tab[0][0]);
    F::add(t0, t0, ca);
    F::mul(t0, t0, tab[0][0]);
    F::add(t0, t0, cb);
    F::sqr(t1, tab[0][1]);
    valid = F::eq(t0, t1);
  }
  el X = tab[0][0], Y = tab[0][1], Z = one;
  ec_dbl_jac<F>(X, Y, Z, ca);          // 2P = (X2 : Y2 : Z2)
  const el X2 = X, Y2 = Y, Z2 = Z;
  {
    el zz, t;
    F::sqr(zz, Z2);
    F::mul(X, tab[0][0], zz);
    F::mul(t, zz, Z2);
    F::mul(Y, tab[0][1], t);
    Z = one;
  }
  for (int j = 1; j < TE; j++) {
    ec_madd_inc<F>(X, Y, Z, X2, Y2);
    tab[j][0] = X;
    tab[j][1] = Y;
    F::mul(zs[j], Z, Z2);
    if (j == 1) cs[1] = zs[1];
    else F::mul(cs[j], cs[j - 1], zs[j]);
  }
  bool bad = F::is0(cs[TE - 1]);
  el zi;
  F::inv(zi, cs[TE - 1]);
  for (int j = TE - 1; j >= 1; j--) {
    el zinv, zz, t;
    if (j > 1) {
      F::mul(zinv, zi, cs[j - 1]);
      F::mul(zi, zi, zs[j]);
    } else {
      zinv = zi;
    }
    F::sqr(zz, zinv);
    F::mul(tab[j][0], tab[j][0], zz);
    F::mul(t, zz, zinv);
    F::mul(tab[j][1], tab[j][1], t);
  }
  uint32_t kw[kScalarWords];
  scalar_load(kw, z, zlen);
  const bool even = (kw[0] & 1) == 0;
  kw[0] |= 1u;
  kw[(8 * zlen) >> 5] |= 1u << ((8 * zlen) & 31);
  const int t = 2 * zlen;
  auto entry = [&](el &x, el &y, int i) {
    const uint32_t v = scalar_bits(kw, 4 * i + 1, 15u);
    const bool neg = v < 8;
    const int idx = neg ? 7 - (int) v : (int) v - 8;
    x = tab[idx][0];
    y = tab[idx][1];
    el ny = F::zero();
    F::sub(ny, ny, y);
    F::cmov(y, ny, neg);
  };
  entry(X, Y, t - 1);                  // the top digit is positive
  Z = one;
  for (int i = t - 2; i >= 0; i--) {
    el x2, y2;
    for (int d = 0; d < WIN; d++) ec_dbl_jac<F>(X, Y, Z, ca);
    entry(x2, y2, i);
    ec_madd_inc<F>(X, Y, Z, x2, y2);
  }
  {
    el sX = X, sY = Y, sZ = Z, ny = F::zero();
    F::sub(ny, ny, tab[0][1]);
    ec_madd_inc<F>(sX, sY, sZ, tab[0][0], ny);
    F::cmov(X, sX, even);
    F::cmov(Y, sY, even);
    F::cmov(Z, sZ, even);
  }
  bad |= F::is0(Z);
  el zinv, zz, ax, ay;
  F::inv(zinv, Z);
  F::sqr(zz, zinv);
  F::mul(ax, X, zz);
  F::mul(zz, zz, zinv);
  F::mul(ay, Y, zz);
  if (!valid) { ax = F::zero(); ay = ax; }
  const bool handled = !valid | !bad;
  if (handled) {
    F::store(out, ax);
    F::store(out + NB, ay);
  }
  return handled;
}

// Fixed-base tables (element_pp_init / element_pp_pow_zn, include/pbc_field.h:591-625 -> element_build_base_table /
// element_pow_base_table, arith/field.c:243-323: 5-bit rows there, products of table entries).  Here 8-bit rows:
// entry (row, w) = [w 2^(8 row)] B for w = 1 .. 255 as an affine point in Montgomery words -- [k] B is one mixed addition
// per scalar byte and one inversion, no doublings.  The table is uniform data in global memory (L2-resident: rows x 255 x
// 2 elements); the lane's digit selects the entry.
constexpr int kPpWin = 8, kPpRowLen = (1 << kPpWin) - 1;
// one table entry per lane: unit u = row * 255 + (w - 1); `flags[u]` = 1 when the entry is O (a base of small order)
template <class F>
PBC_DEV void ec_pp_entry_lane(uint32_t *tab, uint8_t *flags, const uint8_t *base, int zlen, size_t u) {
  typedef typename F::el el;
  const int row = (int) (u / kPpRowLen), w = (int) (u % kPpRowLen) + 1;
  const int NB = F::bytes();
  __attribute__((aligned(16))) uint8_t zb[4 * kScalarWords], o[8 * F::WORDS_EL];      // (records are read and written with word accesses)
  for (int i = 0; i < zlen + 1; i++) zb[i] = 0;
  if (row < zlen) zb[zlen - row] = (uint8_t) w;      // a scalar of zlen + 1 bytes (big-endian): w 2^(8 row)
  ec_mul_lane<F>(o, base, zb, zlen + 1);
  el x, y;
  F::load(x, o);
  F::load(y, o + NB);
  bool inf = true;
  for (int i = 0; i < 2 * NB; i++) inf &= o[i] == 0;
  flags[u] = inf ? 1 : 0;
  F::to_words(tab + u * 2 * F::WORDS_EL, x);
  F::to_words(tab + (u * 2 + 1) * F::WORDS_EL, y);
}
// out = [k] B from the table; false (nothing written) when the lane needs the complete routine
template <class F>
PBC_DEV bool ec_pp_pow_lane(uint8_t *out, const uint32_t *__restrict__ tab, const uint8_t *z, int zlen) {
  typedef typename F::el el;
  const int NB = F::bytes();
  const el one = F::one();
  el X = one, Y = one, Z = one;
  bool inf = true;                     // the accumulator is still O
  for (int row = 0; row < zlen; row++) {
    const uint32_t w = z[zlen - 1 - row];
    const size_t u = (size_t) row * kPpRowLen + (w ? w - 1 : 0);
    const el x2 = F::from_words(tab + u * 2 * F::WORDS_EL), y2 = F::from_words(tab + (u * 2 + 1) * F::WORDS_EL);
    el sX = X, sY = Y, sZ = Z;
    ec_madd_inc<F>(sX, sY, sZ, x2, y2);
    const bool take = w != 0, first = take & inf, add = take & !inf;
    F::cmov(X, sX, add);
    F::cmov(Y, sY, add);
    F::cmov(Z, sZ, add);
    F::cmov(X, x2, first);
    F::cmov(Y, y2, first);
    inf &= !take;
  }
  const bool bad = !inf & F::is0(Z);
  el zinv, zz, ax, ay;
  F::inv(zinv, Z);
  F::sqr(zz, zinv);
  F::mul(ax, X, zz);
  F::mul(zz, zz, zinv);
  F::mul(ay, Y, zz);
  if (inf) { ax = F::zero(); ay = ax; }            // k = 0
  if (!bad) {
    F::store(out, ax);
    F::store(out + NB, ay);
  }
  return !bad;
}

// ---- element_from_hash on G1 (and G2 of the symmetric types) ---------------------------------
// curve_from_hash (ecc/curve.c:455-482): x <- fp_from_hash(data) (arith/montfp.c:440-448 over
// pbc_mpz_from_hash, arith/field.c:643-668: the digest is laid out as H || 0 || H || 1 || ... up to
// the byte length of q, read big-endian and halved while it exceeds q); then x <- x^2 + 1 until
// x^3 + a x + b is a square, y = its square root with the canonical residue ODD
// (element_sgn, montfp.c:457-470), finally the cofactor multiplication (when the curve has one).
// For q = 3 mod 4, sqrt(t) = t^((q+1)/4) and "t is a square" is "that power squares back to t" --
// one power per attempt instead of a Legendre symbol plus a Tonelli run; other fields take one
// power plus a short Tonelli-Shanks tail (fp_sqrt_lane).  Either root will do: the sign rule picks
// the canonical one.  Lanes retry independently under a wave-uniform loop (__all).
template <int N>
static __device__ __noinline__ typename vecN<N>::type fp_pow_sqrt_fn(typename vecN<N>::type va) {
  fp<N> a, r;
  from_vec<N>(a, va);
  fp_set<N>(r, fpk<N>().one);
  for (int i = c_curve.sqrt_bits - 1; i >= 0; i--) {
    fp_sqr<N>(r, r);
    if ((c_curve.sqrt_e[i >> 5] >> (i & 31)) & 1) fp_mul<N>(r, r, a);
  }
  return to_vec<N>(r);
}
// Square root attempt for one lane: y with y^2 = t, or ok = false when t is not a square.
//   q = 3 mod 4: y = t^((q+1)/4), checked by squaring.
//   otherwise Tonelli-Shanks with q - 1 = 2^s t', t' odd: one power w = t^((t'-1)/2) gives the
//   candidate root t w and b = t w^2 = t^t'; the 2-power part is removed in at most s rounds with
//   c = z^t' (z a fixed non-residue).  Control flow is wave-uniform (fixed trip counts, masked updates).
template <int N>
PBC_DEV void fp_sqrt_lane(fp<N> &y, bool &ok, const fp<N> &t) {
  fp<N> w;
  from_vec<N>(w, fp_pow_sqrt_fn<N>(to_vec<N>(t)));
  if (c_curve.sqrt_mode == 0) {
    fp<N> yy;
    fp_sqr<N>(yy, w);
    ok = fp_eq<N>(yy, t);
    y = w;
    return;
  }
  fp<N> one, r, b, c, g, tt;
  fp_set<N>(one, fpk<N>().one);
  fp_mul<N>(r, t, w);
  fp_mul<N>(b, r, w);
  fp_set<N>(c, c_curve.ts_c);
  const int s = c_curve.ts_s;
  int m = s;
  ok = true;
  for (int round = 0; round < s; round++) {
    // least i with b^(2^i) = 1 (i = 0: finished; i >= m: t is not a square)
    int i = 0;
    bool found = fp_eq<N>(b, one);
    tt = b;
    for (int j = 1; j <= s; j++) {
      fp_sqr<N>(tt, tt);
      bool hit = !found & fp_eq<N>(tt, one);
      i = hit ? j : i;
      found |= hit;
    }
    ok &= found & (i < m || i == 0);
    const bool upd = ok & (i > 0) & (i < m);
    g = c;
    for (int j = 0; j < s; j++) {      // g = c^(2^(m-i-1))
      fp_sqr<N>(tt, g);
      fp_cmov<N>(g, tt, upd & (j < m - i - 1));
    }
    fp_mul<N>(tt, r, g);
    fp_cmov<N>(r, tt, upd);
    fp_sqr<N>(g, g);
    fp_cmov<N>(c, g, upd);
    fp_mul<N>(tt, b, g);
    fp_cmov<N>(b, tt, upd);
    m = upd ? i : m;
  }
  ok &= fp_eq<N>(b, one);
  y = r;
}
// z^t' for the smallest non-residue z = 2, 3, ... (single lane, once per parameter set)
template <int N>
PBC_DEV void fp_ts_init(uint32_t *out, const uint32_t *texp, int tbits, const uint32_t *half, int halfbits) {
  fp<N> one, z, chk, c;
  fp_set<N>(one, fpk<N>().one);
  z = one;
  for (;;) {
    fp_add<N>(z, z, one);
    fp_set<N>(chk, fpk<N>().one);
    for (int i = halfbits - 1; i >= 0; i--) {
      fp_sqr<N>(chk, chk);
      if ((half[i >> 5] >> (i & 31)) & 1) fp_mul<N>(chk, chk, z);
    }
    if (!fp_eq<N>(chk, one)) break;
  }
  fp_set<N>(c, fpk<N>().one);
  for (int i = tbits - 1; i >= 0; i--) {
    fp_sqr<N>(c, c);
    if ((texp[i >> 5] >> (i & 31)) & 1) fp_mul<N>(c, c, z);
  }
  for (int k = 0; k < N; k++) out[k] = c.v[k];
}
// pbc_mpz_from_hash (arith/field.c:643-668, called by fp_from_hash, montfp.c:441-449): fbytes bytes = H || ctr || H || ctr+1 ... read as a
// big-endian integer z, then z >>= 1 while z > q; result in Montgomery form
template <int N>
PBC_DEV void fq_from_hash_lane(fp<N> &x, const uint8_t *data, int hlen) {
  const int NB = (int) fpk<N>().fbytes;
  const FpK<N> &K = fpk<N>();
  uint32_t w[N];
#pragma unroll
  for (int i = 0; i < N; i++) w[i] = 0;
  int p = 0;
  uint32_t ctr = 0;
  for (int b = 0; b < NB; b++) {
    uint32_t byte;
    if (p < hlen) byte = data[p++];
    else { byte = ctr++ & 0xff; p = 0; }
    const int bit = 8 * (NB - 1 - b);
#pragma unroll
    for (int i = 0; i < N; i++) w[i] |= (i == (bit >> 5)) ? byte << (bit & 31) : 0u;
  }
  for (int rep = 0; rep < 8; rep++) {
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < N; i++) (void) __builtin_subc(K.p[i], w[i], bw, &bw);
    if (bw) {                          // q - z borrowed: z > q
#pragma unroll
      for (int i = 0; i < N; i++) w[i] = (w[i] >> 1) | (i + 1 < N ? w[i + 1] << 31 : 0u);
    }
  }
  fp<N> t, r2;
#pragma unroll
  for (int i = 0; i < N; i++) t.v[i] = w[i];
  fp_set<N>(r2, K.r2);
  fp_mul<N>(x, t, r2);
}
// The search of curve_from_hash (ecc/curve.c:462-470): the first x of the chain x0, x0^2 + 1, ... with x^3 + a x + b a square,
// and a root y.  Half of all field elements are squares, so a lane needs two tries on average -- but a wave that lets
// every lane walk its own chain in lockstep runs until its SLOWEST lane is done: 7.3 rounds of a 510-bit power for 64
// lanes.  Here the wave's 64 exponentiation slots of a round are dealt to the lanes that are still searching: with P
// of them pending, slot s works on candidate number s / P of pending lane s mod P (operands travel by ds_bpermute, a
// candidate costs a few squarings to reach), and a lane takes the FIRST of its candidates that is a square, as the
// reference does.  32 lanes are left after the first round, 8 after the second, none after the third: 3.1 rounds expected.
template <int N>
struct WordSqrt {                      // the square-root attempt of the search: y with y^2 = t, or ok = false
  static PBC_DEV void attempt(fp<N> &y, bool &ok, const fp<N> &t) { fp_sqrt_lane<N>(y, ok, t); }
};
template <int N, class Sqrt = WordSqrt<N>>
PBC_DEV void g_hash_search(fp<N> &fx, fp<N> &fy, const fp<N> &x0, const fp<N> &ca, const fp<N> &cb) {
  fp<N> one, x = x0;
  fp_set<N>(one, fpk<N>().one);
  fx = x; fy = x;
  bool done = false;
#ifdef PBC_HOSTSIM
  for (int it = 0; it < 256 && !done; it++) {        // one lane per call: its own chain
    fp<N> t, y;
    fp_sqr<N>(t, x);
    fp_add<N>(t, t, ca);
    fp_mul<N>(t, t, x);
    fp_add<N>(t, t, cb);
    bool ok;
    Sqrt::attempt(y, ok, t);
    if (ok) { fx = x; fy = y; done = true; }
    fp_sqr<N>(x, x);
    fp_add<N>(x, x, one);
  }
#else
  const int lane = (int) (threadIdx.x & 63);
  for (int round = 0; round < 200; round++) {
    const uint64_t pend = __ballot(!done);
    if (!pend) break;
    const int P = __popcll(pend), maxoff = 63 / P;
    // slot `lane` -> candidate number off of the pidx-th pending lane
    const int pidx = lane % P, off = lane / P;
    int src = 0, cnt = 0;
    for (int k = 0; k < 64; k++)
      if ((pend >> k) & 1) { src = cnt == pidx ? k : src; cnt++; }
    fp<N> xc, t, y, nx;
#pragma unroll
    for (int w = 0; w < N; w++) xc.v[w] = (uint32_t) __builtin_amdgcn_ds_bpermute(src << 2, (int) x.v[w]);
    for (int k = 0; k < maxoff; k++) {
      fp_sqr<N>(nx, xc);
      fp_add<N>(nx, nx, one);
      fp_cmov<N>(xc, nx, k < off);
    }
    fp_sqr<N>(t, xc);
    fp_add<N>(t, t, ca);
    fp_mul<N>(t, t, xc);
    fp_add<N>(t, t, cb);
    bool ok;
    Sqrt::attempt(y, ok, t);
    const uint64_t okm = __ballot(ok);
    // a pending lane of rank r owns the slots r, r + P, r + 2 P, ...: the first that found a square wins
    const int r = __popcll(pend & ((1ull << lane) - 1ull));
    int win = -1;
    for (int j = maxoff; j >= 0; j--) {
      const int sl = r + j * P;
      win = (sl < 64 && ((okm >> sl) & 1)) ? sl : win;
    }
    const bool got = !done && win >= 0;
    const int from = got ? win : lane;
#pragma unroll
    for (int w = 0; w < N; w++) {
      const uint32_t vx = (uint32_t) __builtin_amdgcn_ds_bpermute(from << 2, (int) xc.v[w]);
      const uint32_t vy = (uint32_t) __builtin_amdgcn_ds_bpermute(from << 2, (int) y.v[w]);
      fx.v[w] = got ? vx : fx.v[w];
      fy.v[w] = got ? vy : fy.v[w];
    }
    // still searching: step over the candidates this round tried for the lane
    const int tried = (63 - r) / P + 1;
    for (int k = 0; k <= maxoff; k++) {
      fp_sqr<N>(nx, x);
      fp_add<N>(nx, nx, one);
      fp_cmov<N>(x, nx, !done && !got && k < tried);
    }
    done |= got;
  }
#endif
}
template <int N>
PBC_DEV void g_from_hash_lane(uint8_t *out, const uint8_t *data, int hlen) {
  const int NB = (int) fpk<N>().fbytes;
  const FpK<N> &K = fpk<N>();
  fp<N> one, ca, cb, x, fx, fy;
  fp_set<N>(one, K.one);
  fp_set<N>(ca, c_curve.a);
  fp_set<N>(cb, c_curve.b);
  fq_from_hash_lane<N>(x, data, hlen);
  g_hash_search<N>(fx, fy, x, ca, cb);
  // canonical y odd
  {
    fp<N> o, c, ny;
#pragma unroll
    for (int i = 0; i < N; i++) o.v[i] = (i == 0);
    fp_mul<N>(c, fy, o);
    fp_neg<N>(ny, fy);
    fp_cmov<N>(fy, ny, ((c.v[0] & 1) == 0) & !fp_is0<N>(fy));
  }
  // [h] (fx, fy): wave-uniform double-and-add over the cofactor (element_mul_mpz, curve.c:477), complete group law
  fp<N> X = fx, Y = fy, Z = one, DX = fx, DY = fy, DZ = one;
  ec_dbl_jac<FqOps<N>>(DX, DY, DZ, ca);
  for (int i = c_curve.cofbits - 2; i >= 0; i--) {
    ec_dbl_jac<FqOps<N>>(X, Y, Z, ca);
    if ((c_curve.cofac[i >> 5] >> (i & 31)) & 1) ec_madd_jac<FqOps<N>>(X, Y, Z, fx, fy, DX, DY, DZ, true);
  }
  fp<N> zi, zi2, ax, ay;
  bool is_inf = fp_is0<N>(Z);
  fp_inv<N>(zi, Z);
  fp_sqr<N>(zi2, zi);
  fp_mul<N>(ax, X, zi2);
  fp_mul<N>(zi2, zi2, zi);
  fp_mul<N>(ay, Y, zi2);
  if (is_inf) {
#pragma unroll
    for (int k = 0; k < N; k++) { ax.v[k] = 0; ay.v[k] = 0; }
  }
  fp_store_be<N>(out, ax);
  fp_store_be<N>(out + NB, ay);
}

// ---- compressed points on E(F_q) ---------------------------------------------------------------
// element_to_bytes_compressed (ecc/curve.c:762-773): x || s with s = 1 when the canonical y is odd
// (element_sign > 0, montfp.c:457-470), else 0.  element_from_bytes_compressed (:800-815): y from
// x by a square root (point_from_x :779-793), negated when its sign disagrees with s.  An x with no
// point above it ("requires a solution to exist") is written as the zero record (O).
template <int N>
PBC_DEV void g_compress_lane(uint8_t *out, const uint8_t *in) {
  const int NB = (int) fpk<N>().fbytes;
  for (int i = 0; i < NB; i++) out[i] = in[i];
  out[NB] = in[2 * NB - 1] & 1;        // y is a canonical residue < q: its parity is its last byte's
}
// element_to_bytes_x_only / element_from_bytes_x_only (ecc/curve.c:821-836): x alone; the point is rebuilt with
// whichever root element_sqrt returns (point_from_x :778-791, no sign fix-up).  For q = 3 mod 4 that root is
// t^((q+1)/4) (element_tonelli, arith/field.c:672-720, with s = 1) and the bytes match the reference; for
// q = 1 mod 4 the reference's root depends on its randomly drawn non-residue (field_gen_nqr :370-376), so it is
// defined up to sign there and this routine returns the root of its own Tonelli-Shanks constants.
template <int N>
PBC_DEV void g_to_x_only_lane(uint8_t *out, const uint8_t *in) {
  const int NB = (int) fpk<N>().fbytes;
  for (int i = 0; i < NB; i++) out[i] = in[i];
}
template <int N>
PBC_DEV void g_decompress_lane(uint8_t *out, const uint8_t *in, bool x_only = false) {
  const int NB = (int) fpk<N>().fbytes;
  fp<N> x, t, y, ny, ca, cb, o, c;
  fp_set<N>(ca, c_curve.a);
  fp_set<N>(cb, c_curve.b);
  fp_load_be<N>(x, in);
  fp_sqr<N>(t, x);
  fp_add<N>(t, t, ca);
  fp_mul<N>(t, t, x);
  fp_add<N>(t, t, cb);
  bool ok;
  fp_sqrt_lane<N>(y, ok, t);
#pragma unroll
  for (int i = 0; i < N; i++) o.v[i] = (i == 0);
  fp_mul<N>(c, y, o);                  // canonical residue: its parity is the sign
  const bool odd = (c.v[0] & 1) != 0, want_odd = x_only ? odd : in[NB] != 0;
  fp_neg<N>(ny, y);
  fp_cmov<N>(y, ny, (odd != want_odd) & !fp_is0<N>(y));
  if (!ok) {
#pragma unroll
    for (int k = 0; k < N; k++) { x.v[k] = 0; y.v[k] = 0; }
  }
  fp_store_be<N>(out, x);
  fp_store_be<N>(out + NB, y);
}

// ---- GT ------------------------------------------------------------------------------------
// Type A: F_q^2
template <int N>
PBC_DEV void a_gt_load(fp2<N> &r, const uint8_t *s) { fp_load_be<N>(r.x, s); fp_load_be<N>(r.y, s + fq_bytes<N>()); }
template <int N>
PBC_DEV void a_gt_store(uint8_t *d, const fp2<N> &a) { fp_store_be<N>(d, a.x); fp_store_be<N>(d + fq_bytes<N>(), a.y); }
template <int N>
PBC_DEV void a_gt_mul_lane(uint8_t *out, const uint8_t *a, const uint8_t *b) {
  fp2<N> x, y;
  a_gt_load<N>(x, a);
  a_gt_load<N>(y, b);
  fi_mul<N>(x, x, y);
  a_gt_store<N>(out, x);
}
template <int N>
PBC_DEV void a_gt_pow_lane(uint8_t *out, const uint8_t *a, const uint8_t *z, int zlen) {
  fp2<N> x, acc, t;
  a_gt_load<N>(x, a);
  fp_set<N>(acc.x, fpk<N>().one);
#pragma unroll
  for (int k = 0; k < N; k++) acc.y.v[k] = 0;
  for (int i = 8 * zlen - 1; i >= 0; i--) {
    fi_sqr<N>(acc, acc);
    fi_mul<N>(t, acc, x);
    bool bit = zr_bit(z, zlen, i) != 0;
    fp_cmov<N>(acc.x, t.x, bit);
    fp_cmov<N>(acc.y, t.y, bit);
  }
  a_gt_store<N>(out, acc);
}
// Types D and G: F_q^k = F_q^d[sqrt(v)], d = 3 / 5
template <int N, int DEG>
PBC_DEV void d_gt_load(typename TypeMNT<N, DEG>::f6 &r, const uint8_t *s) {
  TypeMNT<N, DEG>::f3_load_be(r.x, s);
  TypeMNT<N, DEG>::f3_load_be(r.y, s + DEG * fpk<N>().fbytes);
}
template <int N, int DEG>
PBC_DEV void d_gt_store(uint8_t *d, const typename TypeMNT<N, DEG>::f6 &a) {
  TypeMNT<N, DEG>::f3_store_be(d, a.x);
  TypeMNT<N, DEG>::f3_store_be(d + DEG * fpk<N>().fbytes, a.y);
}
template <int N, int DEG>
PBC_DEV void d_gt_mul_lane(uint8_t *out, const uint8_t *a, const uint8_t *b) {
  typename TypeMNT<N, DEG>::f6 x, y;
  d_gt_load<N, DEG>(x, a);
  d_gt_load<N, DEG>(y, b);
  TypeMNT<N, DEG>::f6_mul(x, x, y);
  d_gt_store<N, DEG>(out, x);
}
template <int N, int DEG>
PBC_DEV void d_gt_pow_lane(uint8_t *out, const uint8_t *a, const uint8_t *z, int zlen) {
  typename TypeMNT<N, DEG>::f6 x, acc, t;
  d_gt_load<N, DEG>(x, a);
  fp<N> one;
  fp_set<N>(one, fpk<N>().one);
  TypeMNT<N, DEG>::f3_set_fq(acc.x, one);
  TypeMNT<N, DEG>::f3_sub(acc.y, acc.x, acc.x);
  for (int i = 8 * zlen - 1; i >= 0; i--) {
    TypeMNT<N, DEG>::f6_sqr(acc, acc);
    TypeMNT<N, DEG>::f6_mul(t, acc, x);
    bool bit = zr_bit(z, zlen, i) != 0;
#pragma unroll
    for (int c = 0; c < DEG; c++) { fp_cmov<N>(acc.x.c[c], t.x.c[c], bit); fp_cmov<N>(acc.y.c[c], t.y.c[c], bit); }
  }
  d_gt_store<N, DEG>(out, acc);
}
// Type F: F_q^12 (private-memory objects)
template <int ND>
__device__ void f_gt_load(typename TypeF<ND>::f12 *r, const uint8_t *s) {
#pragma nounroll
  for (int i = 0; i < 6; i++) TypeF<ND>::g2_load_be(r->c[i], s + 2 * TypeF<ND>::fb() * i);
}
template <int ND>
__device__ void f_gt_store(uint8_t *d, const typename TypeF<ND>::f12 *a) {
#pragma nounroll
  for (int i = 0; i < 6; i++) TypeF<ND>::g2_store_be(d + 2 * TypeF<ND>::fb() * i, a->c[i]);
}
template <int ND>
__device__ void f_gt_mul_lane(uint8_t *out, const uint8_t *a, const uint8_t *b) {
  typename TypeF<ND>::f12 x, y;
  f_gt_load<ND>(&x, a);
  f_gt_load<ND>(&y, b);
  TypeF<ND>::f12_mul(&x, &x, &y);
  f_gt_store<ND>(out, &x);
}
template <int ND>
__device__ void f_gt_pow_lane(uint8_t *out, const uint8_t *a, const uint8_t *z, int zlen) {
  typedef TypeF<ND> T;
  typename T::f12 x, acc, t;
  f_gt_load<ND>(&x, a);
  T::f12_one(&acc);
  for (int i = 8 * zlen - 1; i >= 0; i--) {
    T::f12_sqr(&acc, &acc);
    T::f12_mul(&t, &acc, &x);
    bool bit = zr_bit(z, zlen, i) != 0;
#pragma nounroll
    for (int c = 0; c < 6; c++) {
      typename T::g2 u = acc.c[c], w = t.c[c];
      fp_cmov<ND>(u.x, w.x, bit);
      fp_cmov<ND>(u.y, w.y, bit);
      acc.c[c] = u;
    }
  }
  f_gt_store<ND>(out, &acc);
}

// ---- GT as a field policy (fixed-base powers) -----------------------------------------------------------------------------
// el, WORDS_EL Montgomery words per element, load / store (GT's wire format), mul, one, to_words / from_words
template <int N>
struct GtA {                           // types a, a1: F_q^2
  typedef fp2<N> el;
  static constexpr int NW = N, WORDS_EL = 2 * N;
  static PBC_DEV int bytes() { return 2 * (int) fpk<N>().fbytes; }
  static PBC_DEV void load(el &r, const uint8_t *s) { a_gt_load<N>(r, s); }
  static PBC_DEV void store(uint8_t *d, const el &a) { a_gt_store<N>(d, a); }
  static PBC_DEV void mul(el &r, const el &a, const el &b) { fi_mul<N>(r, a, b); }
  static PBC_DEV void one(el &r) { fp_set<N>(r.x, fpk<N>().one); for (int k = 0; k < N; k++) r.y.v[k] = 0; }
  static PBC_DEV void to_words(uint32_t *w, const el &a) { for (int k = 0; k < N; k++) { w[k] = a.x.v[k]; w[N + k] = a.y.v[k]; } }
  static PBC_DEV void from_words(el &r, const uint32_t *w) { fp_set<N>(r.x, w); fp_set<N>(r.y, w + N); }
};
template <int N>
struct GtE {                           // type e: F_q
  typedef fp<N> el;
  static constexpr int NW = N, WORDS_EL = N;
  static PBC_DEV int bytes() { return (int) fpk<N>().fbytes; }
  static PBC_DEV void load(el &r, const uint8_t *s) { fp_load_be<N>(r, s); }
  static PBC_DEV void store(uint8_t *d, const el &a) { fp_store_be<N>(d, a); }
  static PBC_DEV void mul(el &r, const el &a, const el &b) { fp_mul<N>(r, a, b); }
  static PBC_DEV void one(el &r) { fp_set<N>(r, fpk<N>().one); }
  static PBC_DEV void to_words(uint32_t *w, const el &a) { for (int k = 0; k < N; k++) w[k] = a.v[k]; }
  static PBC_DEV void from_words(el &r, const uint32_t *w) { fp_set<N>(r, w); }
};
template <int N, int DEG>
struct GtD {                           // types d, g: F_q^k = F_q^d[sqrt(v)]
  typedef TypeMNT<N, DEG> T;
  typedef typename T::f6 el;
  static constexpr int NW = N, WORDS_EL = 2 * DEG * N;
  static PBC_DEV int bytes() { return 2 * DEG * (int) fpk<N>().fbytes; }
  static PBC_DEV void load(el &r, const uint8_t *s) { d_gt_load<N, DEG>(r, s); }
  static PBC_DEV void store(uint8_t *d, const el &a) { d_gt_store<N, DEG>(d, a); }
  static PBC_DEV void mul(el &r, const el &a, const el &b) { T::f6_mul(r, a, b); }
  static PBC_DEV void one(el &r) { fp<N> o; fp_set<N>(o, fpk<N>().one); T::f3_set_fq(r.x, o); T::f3_sub(r.y, r.x, r.x); }
  static PBC_DEV void to_words(uint32_t *w, const el &a) {
    for (int i = 0; i < DEG; i++) for (int k = 0; k < N; k++) { w[N * i + k] = a.x.c[i].v[k]; w[N * (DEG + i) + k] = a.y.c[i].v[k]; }
  }
  static PBC_DEV void from_words(el &r, const uint32_t *w) { for (int i = 0; i < DEG; i++) { fp_set<N>(r.x.c[i], w + N * i); fp_set<N>(r.y.c[i], w + N * (DEG + i)); } }
};
template <int ND>
struct GtF {                           // type f: F_q^12 (private-memory objects)
  typedef TypeF<ND> T;
  typedef typename T::f12 el;
  static constexpr int NW = ND, WORDS_EL = 12 * ND;
  static PBC_DEV int bytes() { return 12 * (int) fpk<ND>().fbytes; }
  static PBC_DEV void load(el &r, const uint8_t *s) { f_gt_load<ND>(&r, s); }
  static PBC_DEV void store(uint8_t *d, const el &a) { f_gt_store<ND>(d, &a); }
  static PBC_DEV void mul(el &r, const el &a, const el &b) { T::f12_mul(&r, &a, &b); }
  static PBC_DEV void one(el &r) { T::f12_one(&r); }
  static PBC_DEV void to_words(uint32_t *w, const el &a) {
    for (int i = 0; i < 6; i++) for (int k = 0; k < ND; k++) { w[2 * ND * i + k] = a.c[i].x.v[k]; w[2 * ND * i + ND + k] = a.c[i].y.v[k]; }
  }
  static PBC_DEV void from_words(el &r, const uint32_t *w) { for (int i = 0; i < 6; i++) { fp_set<ND>(r.c[i].x, w + 2 * ND * i); fp_set<ND>(r.c[i].y, w + 2 * ND * i + ND); } }
};
// entry (row, w) = a^(w 2^(8 row)) for w = 0 .. 255 (w = 0: the identity, so that a power is a plain product over the rows);
// unit u = row * 256 + w, one per lane
template <class G>
PBC_DEV void gt_pp_entry_lane(uint32_t *tab, const uint8_t *a, size_t u) {
  typedef typename G::el el;
  const int row = (int) (u >> kPpWin), w = (int) (u & ((1 << kPpWin) - 1));
  el x, acc;
  G::load(x, a);
  for (int i = 0; i < kPpWin * row; i++) G::mul(x, x, x);
  G::one(acc);
  for (int i = kPpWin - 1; i >= 0; i--) {
    G::mul(acc, acc, acc);
    if ((w >> i) & 1) G::mul(acc, acc, x);
  }
  G::to_words(tab + u * G::WORDS_EL, acc);
}
template <class G>
PBC_DEV void gt_pp_pow_lane(uint8_t *out, const uint32_t *__restrict__ tab, const uint8_t *z, int zlen) {
  typedef typename G::el el;
  el acc, t;
  G::from_words(acc, tab + (size_t) z[zlen - 1] * G::WORDS_EL);
  for (int row = 1; row < zlen; row++) {
    G::from_words(t, tab + ((size_t) row * (1 << kPpWin) + z[zlen - 1 - row]) * G::WORDS_EL);
    G::mul(acc, acc, t);
  }
  G::store(out, acc);
}

// element_pow_zn on GT of type f for elements of the cyclotomic subgroup (every pairing value, and every product and
// power of such): the running power stays in the pairing kernels' LDS area, squarings are Granger-Scott squarings
// (pairing_f.cuh f12_cyc_sqr_lds: ~half the work of a general F_q^12 squaring), the scalar is recoded into regular signed
// 4-bit windows as for the curve groups (inversion is the q^6-power: a conjugation), one product per window with an entry
// of the per-lane table a, a^3, ..., a^15.  T is the instantiation the object's pairing kernels use (i-basis / sparse xi:
// the element is mapped into that basis on the way in and back on the way out, as a Miller value is).  Membership in the
// subgroup is TESTED (a^(q^4) a = a^(q^2)): any other element is reported (false, nothing written) for f_gt_pow_lane.
template <class T>
__device__ bool f_gt_pow_cyc_lane(uint8_t *out, const uint8_t *ab, const uint8_t *z, int zlen) {
  constexpr int ND = sizeof(typename T::fq) / 4;
  typedef typename T::f12 f12;
  typedef typename T::g2 g2;
  if constexpr (!(T::kLdsMiller && T::kOneArea && T::kCap >= 9)) {
    return false;
  } else {
    const int fb = T::fb();
    const bool bm1 = c_f.bm1 != 0, xs = c_f.xs_ok != 0;
    f12 a;
    {
      g2 c, cpw;
      T::g2_inv(c, T::fk2(c_f.xc_inv));
      T::g2_zero(cpw);
      fp_set<ND>(cpw.x, fpk<ND>().one);
#pragma nounroll
      for (int i = 0; i < 6; i++) {
        g2 t;
        T::g2_load_be(t, ab + 2 * fb * i);
        if (bm1) fp_mul<ND>(t.y, t.y, T::dk(c_f.cmap));
        if (xs && i) { T::g2_mul(cpw, cpw, c); T::g2_mul(t, t, cpw); }
        a.c[i] = t;
      }
    }
    bool member = c_f.cyc_ok != 0;
    {
      f12 t, u;
      T::f12_qpower(&t, &a, c_f.xpowq2);
      T::f12_qpower(&u, &t, c_f.xpowq2);
      T::f12_mul(&u, &u, &a);
#pragma nounroll
      for (int i = 0; i < 6; i++) member &= T::g2_eq(u.c[i], t.c[i]);
      bool zero = true;                                        // 0 passes the identity; 0^0 = 1 is the generic ladder's business
#pragma nounroll
      for (int i = 0; i < 6; i++) zero &= fp_is0<ND>(a.c[i].x) & fp_is0<ND>(a.c[i].y);
      member &= !zero;
    }
    f12 tab[8], e, ec;
    tab[0] = a;
    T::f12_sqr(&e, &a);
#pragma nounroll
    for (int j = 1; j < 8; j++) T::f12_mul(&tab[j], &tab[j - 1], &e);
    uint32_t kw[kScalarWords];
    scalar_load(kw, z, zlen);
    const bool even = (kw[0] & 1) == 0;
    kw[0] |= 1u;
    kw[(8 * zlen) >> 5] |= 1u << ((8 * zlen) & 31);
    auto entry = [&](int i) {          // e <- the table entry of window i, conjugated (odd coefficients negated: pairing_f.cuh f12_conj) for a negative digit
      const uint32_t v = scalar_bits(kw, 4 * i + 1, 15u);
      const bool neg = v < 8;
      const int idx = neg ? 7 - (int) v : (int) v - 8;
#if PBC_F_NO_CONJ
      e = tab[idx];
      T::f12_conj(&ec, &e);
#pragma nounroll
      for (int c = 0; c < 6; c++) {
        g2 p = e.c[c], q = ec.c[c];
        fp_cmov<ND>(p.x, q.x, neg);
        fp_cmov<ND>(p.y, q.y, neg);
        e.c[c] = p;
      }
      return;
#endif
#pragma nounroll
      for (int c = 0; c < 6; c++) {
        g2 p = tab[idx].c[c];
        if (c & 1) {
          g2 q;
          T::g2_neg(q, p);
          fp_cmov<ND>(p.x, q.x, neg);
          fp_cmov<ND>(p.y, q.y, neg);
        }
        e.c[c] = p;
      }
    };
    const int t = 2 * zlen;
    entry(t - 1);
    T::f12_lds_import(&e, 0);
    for (int i = t - 2; i >= 0; i--) {
      for (int d = 0; d < 4; d++) T::f12_cyc_sqr_lds();
      entry(i);
      T::f12_mul_lds(0, &e);
    }
    f12 r, r2;
    T::f12_lds_export(&r, 0);
    T::f12_conj(&ec, &a);                          // even k: a^k = a^(k + 1) / a
    T::f12_mul_lds(0, &ec);
    T::f12_lds_export(&r2, 0);
    {
      g2 ci = T::fk2(c_f.xc_inv), cpw;
      T::g2_zero(cpw);
      fp_set<ND>(cpw.x, fpk<ND>().one);
#pragma nounroll
      for (int i = 0; i < 6; i++) {
        g2 p = r.c[i], q = r2.c[i];
        fp_cmov<ND>(p.x, q.x, even);
        fp_cmov<ND>(p.y, q.y, even);
        if (xs && i) { T::g2_mul(cpw, cpw, ci); T::g2_mul(p, p, cpw); }
        if (bm1) fp_mul<ND>(p.y, p.y, T::dk(c_f.cinv));
        if (member) T::g2_store_be(out + 2 * fb * i, p);
      }
    }
    return member;
  }
}

// ---- pairing->finalpow (include/pbc_pairing.h:41; a_finalpow a_param.c:1420-1429, cc_finalpow d_param.c:566-568,
// g_finalpow g_param.c:1162-1164, f_finalpow f_param.c:285-287, e_finalpow e_param.c:828-830): the final exponentiation
// alone, on an element of GT's underlying field in GT's wire format (the consumers are gt_random / gt_from_hash,
// ecc/pairing.c:121,127).  The pairing kernels' own final-exponentiation routines, one element per lane.
template <int N>
PBC_DEV void a_finalpow_lane(uint8_t *out, const uint8_t *a) {
  fp2<N> x, r;
  a_gt_load<N>(x, a);
  a_final_exp<N>(r, x);
  a_gt_store<N>(out, r);
}
template <int N, int DEG>
PBC_DEV void d_finalpow_lane(uint8_t *out, const uint8_t *a) {
  typename TypeMNT<N, DEG>::f6 x, r;
  d_gt_load<N, DEG>(x, a);
  TypeMNT<N, DEG>::d_final_exp(r, x);
  d_gt_store<N, DEG>(out, r);
}
template <int ND>
__device__ void f_finalpow_lane(uint8_t *out, const uint8_t *a) {
  typename TypeF<ND>::f12 x;
  f_gt_load<ND>(&x, a);
  TypeF<ND>::f_final_exp(&x);
  f_gt_store<ND>(out, &x);
}
template <int N>
PBC_DEV void e_finalpow_lane(uint8_t *out, const uint8_t *a) {
  fp<N> x, r;
  fp_load_be<N>(x, a);
  e_pow<N>(r, x, c_e.phik, c_e.phikbits);
  fp_store_be<N>(out, r);
}

// ---- square roots, element_from_hash and compressed points on the twists (G2 of types d, g, f) -----------
// The reference takes square roots in F_q^d with a randomised Cantor-Zassenhaus step (polymod_sqrt,
// arith/poly.c:634-700) and in F_q^2 with the norm formula (fq_sqrt, fieldquadratic.c:357-420); both callers
// below normalise the sign afterwards, so any root serves.  Here: Tonelli-Shanks in the extension itself,
// |F*| = 2^s T with T odd, from the non-residue the tower is built on; wave-uniform control flow as in
// fp_sqrt_lane.
struct ExtSqrtK {
  uint32_t e[24], t[24];               // (T - 1)/2 and T
  int ebits, tbits, s;
  uint32_t c[40];                      // z^T (Montgomery words, coefficient-major); derived on the device
};
static_assert(sizeof(ExtSqrtK) <= KOFF_OPT - KOFF_XS, "constant block layout");
#define c_xs (pbc::kconst<pbc::ExtSqrtK, pbc::KOFF_XS>())

template <class F>
PBC_DEV void ext_pow(typename F::el &r, const typename F::el &a, const uint32_t *e, int bits) {
  r = F::one();
  for (int i = bits - 1; i >= 0; i--) {
    F::sqr(r, r);
    if ((e[i >> 5] >> (i & 31)) & 1) F::mul(r, r, a);
  }
}
template <class F>
PBC_DEV void ext_ts_init(uint32_t *out) {
  typename F::el c;
  ext_pow<F>(c, F::nonresidue(), c_xs.t, c_xs.tbits);
  F::to_words(out, c);
}
template <class F>
PBC_DEV void ext_sqrt_lane(typename F::el &y, bool &ok, const typename F::el &t) {
  typedef typename F::el el;
  el w, r, b, c, g, tt;
  const el one = F::one();
  ext_pow<F>(w, t, c_xs.e, c_xs.ebits);
  F::mul(r, t, w);
  F::mul(b, r, w);
  c = F::from_words(c_xs.c);
  const int s = c_xs.s;
  int m = s;
  ok = true;
  for (int round = 0; round < s; round++) {
    int i = 0;
    bool found = F::eq(b, one);
    tt = b;
    for (int j = 1; j <= s; j++) {
      F::sqr(tt, tt);
      bool hit = !found & F::eq(tt, one);
      i = hit ? j : i;
      found |= hit;
    }
    ok &= found & (i < m || i == 0);
    const bool upd = ok & (i > 0) & (i < m);
    g = c;
    for (int j = 0; j < s; j++) {
      F::sqr(tt, g);
      F::cmov(g, tt, upd & (j < m - i - 1));
    }
    F::mul(tt, r, g);
    F::cmov(r, tt, upd);
    F::sqr(g, g);
    F::cmov(c, g, upd);
    F::mul(tt, b, g);
    F::cmov(b, tt, upd);
    m = upd ? i : m;
  }
  ok &= F::eq(b, one);
  y = r;
}
template <int N, int DEG>
PBC_DEV void FdOps<N, DEG>::sqrt_ref(typename FdOps<N, DEG>::el &y, bool &ok, const typename FdOps<N, DEG>::el &t) {
  ext_sqrt_lane<FdOps<N, DEG>>(y, ok, t);
}
// element_to_bytes_x_only / element_from_bytes_x_only (ecc/curve.c:821-836) on E'(K): x alone, y = element_sqrt
// as it comes.  Type f with q = 3 mod 4 reproduces the reference's root (fq_sqrt is a formula over F_q roots);
// elsewhere the reference's root depends on random choices (polymod_sqrt, element_tonelli's non-residue) and the
// result agrees with it up to sign.
template <class F>
PBC_DEV void g2_from_x_lane(uint8_t *out, const uint8_t *in) {
  typedef typename F::el el;
  el x, t, y;
  F::load(x, in);
  F::sqr(t, x);
  F::add(t, t, F::curve_a());
  F::mul(t, t, x);
  F::add(t, t, F::curve_b());
  bool ok;
  F::sqrt_ref(y, ok, t);
  if (!ok) { x = F::zero(); y = F::zero(); }
  F::store(out, x);
  F::store(out + F::bytes(), y);
}
// curve_from_hash (ecc/curve.c:455-482) on E'(K): x from the digest, x <- x^2 + 1 until x^3 + a x + b is a
// square, y the root with non-negative sign.  The twists are initialised without a cofactor
// (d_param.c:1057, f_param.c:383, g_param.c:1319), so no multiplication follows.
template <class F>
PBC_DEV void g2_from_hash_lane(uint8_t *out, const uint8_t *data, int hlen) {
  typedef typename F::el el;
  const el one = F::one(), ca = F::curve_a(), cb = F::curve_b();
  el x = F::hash_x(data, hlen), fx = x, fy = x;
  bool done = false;
  for (int it = 0; it < 256; it++) {
    el t, y;
    F::sqr(t, x);
    F::add(t, t, ca);
    F::mul(t, t, x);
    F::add(t, t, cb);
    bool ok;
    ext_sqrt_lane<F>(y, ok, t);
    ok &= !done;
    F::cmov(fx, x, ok);
    F::cmov(fy, y, ok);
    done |= ok;
    if (__all(done)) break;
    F::sqr(x, x);
    F::add(x, x, one);
  }
  {
    el ny = F::zero();
    F::sub(ny, ny, fy);
    F::cmov(fy, ny, F::is_negative(fy));
  }
  F::store(out, fx);
  F::store(out + F::bytes(), fy);
}
// element_to_bytes_compressed / element_from_bytes_compressed (ecc/curve.c:762-815) on E'(K): x || s with
// s = 1 when element_sign(y) > 0
template <class F>
PBC_DEV void g2_compress_lane(uint8_t *out, const uint8_t *in) {
  typedef typename F::el el;
  const int NB = F::bytes();
  el y;
  F::load(y, in + NB);
  for (int i = 0; i < NB; i++) out[i] = in[i];
  out[NB] = (!F::is_negative(y) & !F::is0(y)) ? 1 : 0;
}
template <class F>
PBC_DEV void g2_decompress_lane(uint8_t *out, const uint8_t *in) {
  typedef typename F::el el;
  const int NB = F::bytes();
  el x, t, y, ny = F::zero();
  F::load(x, in);
  F::sqr(t, x);
  F::add(t, t, F::curve_a());
  F::mul(t, t, x);
  F::add(t, t, F::curve_b());
  bool ok;
  ext_sqrt_lane<F>(y, ok, t);
  // :806-810: s != 0 wants sign(y) >= 0, s == 0 wants sign(y) <= 0
  const bool neg = F::is_negative(y), want_pos = in[NB] != 0;
  F::sub(ny, ny, y);
  F::cmov(y, ny, (neg == want_pos) & !F::is0(y));
  if (!ok) { x = F::zero(); y = F::zero(); }
  F::store(out, x);
  F::store(out + NB, y);
}

}  // namespace pbc
