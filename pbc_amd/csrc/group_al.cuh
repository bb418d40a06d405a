// group_al.cuh -- element_mul_zn on E(F_q): y^2 = x^3 + x and element_pow_zn on GT = F_q^2 for the 512-bit type a field,
// on the LIMB-FORM arithmetic of pairing_al.cuh (18 limbs of 29 bits, redundant representation, no conditional
// subtractions; bounds tracked by the host mirror).
//
// Reference: element_mul_zn / element_pow_zn (include/pbc_field.h:311, :374) -> generic_pow_mpz (arith/field.c:113-126:
// sliding window over the affine group law curve_mul / curve_double, ecc/curve.c:102-207, one mpz_invert per step) and,
// on GT, over fi_mul / fi_sqr (arith/fieldquadratic.c:425-477).  Group elements are unique, so any addition chain gives
// the same bytes.  Here:
//   * G1 / G2 (the same curve for type a): a REGULAR signed fixed-window ladder, w = 4.  For an odd scalar k < 2^n the digits
//     d_i = 2 ((k >> (4 i + 1)) & 15) - 15 (bit n of k taken as 1) are all odd, k = sum d_i 16^i, so every window is four
//     Jacobian doublings and ONE mixed addition of +-T[|d_i| >> 1] from a per-lane table T = {P, 3P, ..., 15P} of affine
//     points -- lanes hold different scalars, the instruction stream stays wave-uniform (the per-lane part is the table
//     index and a sign).  An even scalar runs as k + 1 with one conditional subtraction of P at the end.  The table is
//     built without a separate inversion for 2P: on the isomorphic curve E': y^2 = x^3 + Z2^4 x the point 2P = (X2 : Y2 :
//     Z2) is affine, the odd multiples follow by mixed additions there (additions do not involve the curve coefficient)
//     and go back to E by Z <- Z Z2; one batched inversion makes them affine.  Two inversions per scalar multiplication
//     (safegcd) against the ~190 of the reference's affine ladder.
//   * The step formulas are the incomplete ones of the Miller loop (pairing_al.cuh double_step / add_step without the
//     line): an addition that meets R = +-T or R = O has H = 0 and leaves Z = 0 for good.  The lane routine reports
//     "Z = 0 at the end" (the true result O, a point of small order, an exceptional scalar) and the caller runs the
//     complete word-form routine of group_ops.cuh (ec_mul_lane) for those lanes only -- a second, nearly empty launch.
//   * GT: elements of norm 1 (every pairing value, every power and product of them) are powered with the Lucas ladder of
//     the final exponentiation (lucas_odd, ecc/a_param.c:226-283): V_k(2 Re a) by one product and one squaring in F_q per
//     bit with per-lane selects, Im a^k = -(2 V_(k+1) - P V_k) / (4 Im a).  Elements of other norm are reported for the
//     generic routine (a_gt_pow_lane).
#pragma once
#include "pairing_al.cuh"
#include "group_ops.cuh"

namespace pbc {

template <int N>
struct GAL {
  typedef AL<N> A;
  typedef typename A::el el;
  typedef typename A::jacl jacl;
  static constexpr int L = A::L;
  static constexpr int WIN = 4, TE = 1 << (WIN - 1);       // window width, table entries (odd multiples 1, 3, ..., 2 TE - 1)
  enum { SLOT_FX = A::SLOT_FX, SLOT_FY = A::SLOT_FY, SLOT_Z = A::SLOT_Z, SLOT_ZZ = A::SLOT_ZZ };
  enum { K2 = A::K2, K4 = A::K4, K8 = A::K8, K12 = A::K12, K16 = A::K16 };

  static PBC_DEV void sel(el &r, const el &a, bool take) {       // r = take ? a : r
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = take ? a.l[i] : r.l[i];
    AL_HS(r.hs_u = a.hs_u > r.hs_u ? a.hs_u : r.hs_u; r.hs_B = a.hs_B > r.hs_B ? a.hs_B : r.hs_B;)
  }
  static PBC_DEV el one_el() {
    fp<N> w;
    el r;
    fp_set<N>(w, fpk<N>().one);
    A::to_el(r, w);
    return r;
  }

  // V <- 2V on y^2 = x^3 + x.  V = (X, Y) in registers (almost normalised, B <= 14), Z and Z^2 in their LDS slots (P-class).
  // The point part of AL::double_step: 3 products + 6 squarings.
  static PBC_DEV void ec_dbl(jacl &V) {
    el XX, YY, M, t0, t1, Z3, S1, W;
    A::sqr(XX, V.X);
    A::lds_get(t0, SLOT_ZZ);
    A::sqr(t0, t0);                    // Z^4
    A::template shl<1>(M, XX);
    A::add(M, M, XX);
    A::add(M, M, t0);                  // M = 3X^2 + Z^4: u 4, B 5
    A::norm(M, M);
    A::sqr(YY, V.Y);
    A::template shl<1>(t1, V.Y);       // u 2
    A::muls(Z3, t1, SLOT_Z);           // Z3 = 2YZ
    A::lds_put(SLOT_Z, Z3);
    A::sqr(Z3, Z3);
    A::lds_put(SLOT_ZZ, Z3);
    A::template shl<1>(t1, V.X);       // u 2
    A::mul(S1, YY, t1);                // 2XY^2
    A::sqr(t0, M);
    A::template shl<2>(t1, S1);        // 8XY^2: u 4, B 6
    A::subk(t0, t0, t1, K8);           // u 6, B 9.5
    A::norm(V.X, t0);                  // X3 = M^2 - 2S
    A::template shl<1>(t1, S1);        // S = 4XY^2: u 2, B 3
    A::subk(W, t1, V.X, K12);          // S - X3: u 5, B 15
    A::norm(W, W);
    A::mul(t0, M, W);
    A::sqr(t1, YY);                    // Y^4
    A::template shl<3>(t1, t1);        // u 8, B 12
    A::norm(t1, t1);
    A::subk(t0, t0, t1, K12);          // u 4, B 13.1
    A::norm(V.Y, t0);                  // Y3 = M (S - X3) - 8Y^4
  }
  // V <- V + (x2, y2), mixed.  The point part of AL::add_step: 8 products + 3 squarings.  x2, y2: limbs <= 2^29 + 6, B <= 14.
  // Incomplete: V = +-(x2, y2) or V = O give H = 0 and Z3 = 0, which later steps keep.
  static PBC_DEV void ec_madd(jacl &V, const el &x2, const el &y2) {
    el H, R, Z3, HH, HHH, t0, t1;
    A::muls(t0, x2, SLOT_ZZ);
    A::subk(H, t0, V.X, K16);
    A::norm(H, H);                     // B 17.5
    A::lds_get(t0, SLOT_Z);
    A::muls(t0, t0, SLOT_ZZ);
    A::mul(t0, y2, t0);
    A::subk(R, t0, V.Y, K16);
    A::norm(R, R);                     // B 17.5
    A::muls(Z3, H, SLOT_Z);
    A::sqr(HH, H);
    A::mul(HHH, HH, H);
    A::mul(t0, V.X, HH);               // X1 H^2
    A::sqr(t1, R);
    A::subk(t1, t1, HHH, K2);
    A::norm(t1, t1);                   // B 3.5
    A::template shl<1>(HH, t0);        // u 2, B 3
    A::subk(t1, t1, HH, K4);
    A::norm(t1, t1);                   // X3 = R^2 - H^3 - 2 X1 H^2, B 7.5
    A::subk(t0, t0, t1, K16);
    A::norm(t0, t0);                   // B 17.5
    A::mul(t0, R, t0);
    A::mul(HHH, V.Y, HHH);
    A::subk(t0, t0, HHH, K2);
    A::norm(V.Y, t0);                  // Y3 = R (X1 H^2 - X3) - Y1 H^3, B 3.5
    V.X = t1;
    A::lds_put(SLOT_Z, Z3);
    A::sqr(Z3, Z3);
    A::lds_put(SLOT_ZZ, Z3);
  }
  // 1 / a for a P-class a (the one word-form excursion: safegcd works on words)
  static PBC_DEV void inv(el &r, const el &a) {
    fp<N> w;
    A::to_words(w, a);
    fp_inv<N>(w, w);
    A::to_el(r, w);
  }
  // a P-class value is zero mod q
  static PBC_DEV bool is0(const el &a) {
    fp<N> w;
    A::to_words(w, a);
    return fp_is0<N>(w);
  }

  // The scalar, read ONCE per lane (word loads when the records allow it: over PCIe a byte load is a transaction), as
  // little-endian words in private memory with one zero word above.  KW words hold any Z_r element of these parameter sets.
  static constexpr int KW = N + 2;
  static PBC_DEV void load_scalar(uint32_t *kw, const uint8_t *z, int zlen) {
    for (int w = 0; w < KW; w++) kw[w] = 0;
    if ((zlen & 3) == 0 && (reinterpret_cast<uintptr_t>(z) & 3) == 0) {
      const uint32_t *zw = reinterpret_cast<const uint32_t *>(z);
      for (int w = 0; w < zlen / 4; w++) kw[w] = __builtin_bswap32(zw[zlen / 4 - 1 - w]);
    } else {
      for (int i = 0; i < zlen; i++) kw[i >> 2] |= (uint32_t) z[zlen - 1 - i] << (8 * (i & 3));
    }
  }
  static PBC_DEV uint32_t scalar_bits(const uint32_t *kw, int pos, uint32_t mask) {
    const uint64_t pair = ((uint64_t) kw[(pos >> 5) + 1] << 32) | kw[pos >> 5];
    return (uint32_t) (pair >> (pos & 31)) & mask;
  }

  // out = [k] P.  Returns false when the lane needs the complete routine (the result is O or the ladder met an exceptional
  // addition); nothing is written then.  An off-curve P is O (curve_from_bytes): zeros, handled.
  static PBC_DEV bool gmul_lane(uint8_t *out, const uint8_t *in, const uint8_t *z, int zlen) {
    constexpr int NB = 4 * N;
    el tab[TE][2];                     // per-lane table of affine odd multiples (private memory, indexed by the lane's digit)
    el Px, Py;
    bool valid;
    {
      fp<N> x, y;
      fp_load_be<N>(x, in);
      fp_load_be<N>(y, in + NB);
      valid = a_on_curve<N>(x, y);
      A::to_el(Px, x);
      A::to_el(Py, y);
    }
    const el one = one_el();
    jacl V;
    el Z2;
    // 2P = (X2 : Y2 : Z2)
    V.X = Px;
    V.Y = Py;
    A::lds_put(SLOT_Z, one);
    A::lds_put(SLOT_ZZ, one);
    ec_dbl(V);
    const el X2 = V.X, Y2 = V.Y;
    A::lds_get(Z2, SLOT_Z);
    {
      // phi(P) = (x Z2^2, y Z2^3) on E', where phi(2P) = (X2, Y2) is affine; then phi((2j + 1) P) = phi((2j - 1) P) + phi(2P)
      el zz, t;
      A::lds_get(zz, SLOT_ZZ);
      A::mul(V.X, Px, zz);
      A::mul(t, zz, Z2);
      A::mul(V.Y, Py, t);
      A::lds_put(SLOT_Z, one);
      A::lds_put(SLOT_ZZ, one);
    }
    tab[0][0] = Px;
    tab[0][1] = Py;
    el zs[TE], cs[TE];                 // Z of entry j on E; prefix products of those
    for (int j = 1; j < TE; j++) {
      el zj;
      ec_madd(V, X2, Y2);
      tab[j][0] = V.X;
      tab[j][1] = V.Y;
      A::lds_get(zj, SLOT_Z);
      A::mul(zs[j], zj, Z2);           // back on E: Z <- Z' Z2
      if (j == 1) cs[1] = zs[1];
      else A::mul(cs[j], cs[j - 1], zs[j]);
    }
    bool bad = is0(cs[TE - 1]);        // some odd multiple (or 2P) is O: a point of small order
    el zi;
    inv(zi, cs[TE - 1]);
    for (int j = TE - 1; j >= 1; j--) {
      el zinv, zz, t, X, Y;
      if (j > 1) {
        A::mul(zinv, zi, cs[j - 1]);
        A::mul(zi, zi, zs[j]);
      } else {
        zinv = zi;
      }
      A::sqr(zz, zinv);
      X = tab[j][0];
      Y = tab[j][1];
      AL_HS(A::hs_set(X, A::U_ALMOST, 8.0); A::hs_set(Y, A::U_ALMOST, 8.0);)
      A::mul(tab[j][0], X, zz);
      A::mul(t, zz, zinv);
      A::mul(tab[j][1], Y, t);
    }
    // the ladder over k' = k | 1, with the bit above the scalar's top byte set (the recoding's bit n)
    const int t = 2 * zlen;            // digits: 8 zlen bits in windows of four
    uint32_t kw[KW];
    load_scalar(kw, z, zlen);
    const bool even = (kw[0] & 1) == 0;
    kw[0] |= 1u;
    kw[(8 * zlen) >> 5] |= 1u << ((8 * zlen) & 31);
    auto digit = [&](int i, int &idx, bool &neg) {
      const uint32_t v = scalar_bits(kw, 4 * i + 1, 15u);
      neg = v < 8;
      idx = neg ? 7 - (int) v : (int) v - 8;
    };
    auto entry = [&](el &x, el &y, int idx, bool neg) {
      x = tab[idx][0];
      y = tab[idx][1];
      AL_HS(A::hs_set(x, A::U_STRICT, 1.5); A::hs_set(y, A::U_STRICT, 1.5);)
      el ny;
      A::negk(ny, y, K2);              // u 2, B 2
      A::norm(ny, ny);
      AL_HS(y.hs_u = ny.hs_u; y.hs_B = ny.hs_B;)
      sel(y, ny, neg);
    };
    {
      int idx;
      bool neg;
      digit(t - 1, idx, neg);          // the top digit is positive (bit n is set)
      entry(V.X, V.Y, idx, false);
      A::lds_put(SLOT_Z, one);
      A::lds_put(SLOT_ZZ, one);
    }
    for (int i = t - 2; i >= 0; i--) {
      int idx;
      bool neg;
      el x2, y2;
      if ((i & 1) == 0) pbc_fair_tick<PBC_A_FAIR_BIT>();
      for (int d = 0; d < WIN; d++) ec_dbl(V);
      digit(i, idx, neg);
      entry(x2, y2, idx, neg);
      ec_madd(V, x2, y2);
    }
    {
      // even k: [k] P = [k + 1] P - P
      el sX = V.X, sY = V.Y, sZ, sZZ, px, ny;
      A::lds_get(sZ, SLOT_Z);
      A::lds_get(sZZ, SLOT_ZZ);
      px = tab[0][0];                  // P again, from the table (not kept in registers across the ladder)
      ny = tab[0][1];
      AL_HS(A::hs_set(px, A::U_STRICT, 1.0); A::hs_set(ny, A::U_STRICT, 1.0);)
      A::negk(ny, ny, K2);
      A::norm(ny, ny);
      ec_madd(V, px, ny);
      sel(sX, V.X, even);
      sel(sY, V.Y, even);
      V.X = sX;
      V.Y = sY;
      el nz, nzz;
      A::lds_get(nz, SLOT_Z);
      A::lds_get(nzz, SLOT_ZZ);
      sel(sZ, nz, even);
      sel(sZZ, nzz, even);
      A::lds_put(SLOT_Z, sZ);
      A::lds_put(SLOT_ZZ, sZZ);
    }
    // to affine
    el Zf, zinv, zz, t3, ax, ay;
    A::lds_get(Zf, SLOT_Z);
    bad |= is0(Zf);
    inv(zinv, Zf);
    A::sqr(zz, zinv);
    A::mul(ax, V.X, zz);
    A::mul(t3, zz, zinv);
    A::mul(ay, V.Y, t3);
    fp<N> x, y;
    A::to_words(x, ax);
    A::to_words(y, ay);
    if (!valid) {
#pragma unroll
      for (int k = 0; k < N; k++) { x.v[k] = 0; y.v[k] = 0; }
    }
    const bool handled = !valid | !bad;
    if (handled) {
      fp_store_be<N>(out, x);
      fp_store_be<N>(out + NB, y);
    }
    return handled;
  }

  // ---- element_pow2_zn / element_pow3_zn (field_pow2 / field_pow3 of the curve group, arith/field.c generic_pow2/3) ---------
  // out = [k_0] P_0 + ... + [k_{KB-1}] P_{KB-1} with ONE accumulator: the ladder above on KB tables -- four doublings and KB
  // mixed additions per window, where the composition of single-base ladders doubles 4 KB times (per window: 4 x 9 + 11 KB
  // products instead of KB (4 x 9 + 11)).  Base j's record: a[j] + idx astride, its scalar: z[j] + idx zstride.  Even
  // scalars run as k + 1 with a conditional subtraction at the end, as above.  Returns false -- nothing written -- when
  // the lane needs the complete routine: a base off the curve, an exceptional addition anywhere (equal or opposite bases
  // meet in the accumulator, a point of small order, a zero scalar), or the result O; all of these leave Z = 0.
  template <int KB>
  static PBC_DEV bool gmulk_lane(uint8_t *out, const uint8_t *const *in, const uint8_t *const *z, int zlen) {
    constexpr int NB = 4 * N;
    el tab[KB][TE][2];                 // per-lane tables of affine odd multiples (private memory)
    uint32_t kw[KB][KW];
    bool even[KB];
    const el one = one_el();
    jacl V;
    bool valid = true, bad = false;
#pragma unroll 1
    for (int b = 0; b < KB; b++) {
      el Px, Py, Z2;
      {
        fp<N> x, y;
        fp_load_be<N>(x, in[b]);
        fp_load_be<N>(y, in[b] + NB);
        valid &= a_on_curve<N>(x, y);
        A::to_el(Px, x);
        A::to_el(Py, y);
      }
      // 2P = (X2 : Y2 : Z2); phi(P) = (x Z2^2, y Z2^3) on E', where phi(2P) is affine; phi((2j + 1) P) = phi((2j - 1) P) + phi(2P)
      V.X = Px;
      V.Y = Py;
      A::lds_put(SLOT_Z, one);
      A::lds_put(SLOT_ZZ, one);
      ec_dbl(V);
      const el X2 = V.X, Y2 = V.Y;
      A::lds_get(Z2, SLOT_Z);
      {
        el zz, t;
        A::lds_get(zz, SLOT_ZZ);
        A::mul(V.X, Px, zz);
        A::mul(t, zz, Z2);
        A::mul(V.Y, Py, t);
        A::lds_put(SLOT_Z, one);
        A::lds_put(SLOT_ZZ, one);
      }
      tab[b][0][0] = Px;
      tab[b][0][1] = Py;
      el zs[TE], cs[TE];               // Z of entry j on E; prefix products of those
      for (int j = 1; j < TE; j++) {
        el zj;
        ec_madd(V, X2, Y2);
        tab[b][j][0] = V.X;
        tab[b][j][1] = V.Y;
        A::lds_get(zj, SLOT_Z);
        A::mul(zs[j], zj, Z2);         // back on E: Z <- Z' Z2
        if (j == 1) cs[1] = zs[1];
        else A::mul(cs[j], cs[j - 1], zs[j]);
      }
      bad |= is0(cs[TE - 1]);          // some odd multiple (or 2P) is O: a point of small order
      el zi;
      inv(zi, cs[TE - 1]);
      for (int j = TE - 1; j >= 1; j--) {
        el zinv, zz, t, X, Y;
        if (j > 1) {
          A::mul(zinv, zi, cs[j - 1]);
          A::mul(zi, zi, zs[j]);
        } else {
          zinv = zi;
        }
        A::sqr(zz, zinv);
        X = tab[b][j][0];
        Y = tab[b][j][1];
        AL_HS(A::hs_set(X, A::U_ALMOST, 8.0); A::hs_set(Y, A::U_ALMOST, 8.0);)
        A::mul(tab[b][j][0], X, zz);
        A::mul(t, zz, zinv);
        A::mul(tab[b][j][1], Y, t);
      }
      // k' = k | 1, with the bit above the scalar's top byte set (the recoding's bit n)
      load_scalar(kw[b], z[b], zlen);
      even[b] = (kw[b][0] & 1) == 0;
      kw[b][0] |= 1u;
      kw[b][(8 * zlen) >> 5] |= 1u << ((8 * zlen) & 31);
    }
    const int t = 2 * zlen;            // digits: 8 zlen bits in windows of four
    auto entry = [&](el &x, el &y, int b, int i, bool top) {
      const uint32_t v = scalar_bits(kw[b], 4 * i + 1, 15u);
      const bool neg = !top && v < 8;  // (the top digit is positive: bit n is set)
      const int idx = v < 8 ? 7 - (int) v : (int) v - 8;
      x = tab[b][idx][0];
      y = tab[b][idx][1];
      AL_HS(A::hs_set(x, A::U_STRICT, 1.5); A::hs_set(y, A::U_STRICT, 1.5);)
      el ny;
      A::negk(ny, y, K2);              // u 2, B 2
      A::norm(ny, ny);
      AL_HS(y.hs_u = ny.hs_u; y.hs_B = ny.hs_B;)
      sel(y, ny, neg);
    };
    entry(V.X, V.Y, 0, t - 1, true);
    A::lds_put(SLOT_Z, one);
    A::lds_put(SLOT_ZZ, one);
#pragma unroll 1
    for (int b = 1; b < KB; b++) {
      el x2, y2;
      entry(x2, y2, b, t - 1, true);
      ec_madd(V, x2, y2);
    }
    for (int i = t - 2; i >= 0; i--) {
      if ((i & 1) == 0) pbc_fair_tick<PBC_A_FAIR_BIT>();
      for (int d = 0; d < WIN; d++) ec_dbl(V);
#pragma unroll 1
      for (int b = 0; b < KB; b++) {
        el x2, y2;
        entry(x2, y2, b, i, false);
        ec_madd(V, x2, y2);
      }
    }
#pragma unroll 1
    for (int b = 0; b < KB; b++) {
      // even k_b: the sum so far holds [k_b + 1] P_b; take P_b off
      el sX = V.X, sY = V.Y, sZ, sZZ, px, ny;
      A::lds_get(sZ, SLOT_Z);
      A::lds_get(sZZ, SLOT_ZZ);
      px = tab[b][0][0];
      ny = tab[b][0][1];
      AL_HS(A::hs_set(px, A::U_STRICT, 1.0); A::hs_set(ny, A::U_STRICT, 1.0);)
      A::negk(ny, ny, K2);
      A::norm(ny, ny);
      ec_madd(V, px, ny);
      sel(sX, V.X, even[b]);
      sel(sY, V.Y, even[b]);
      V.X = sX;
      V.Y = sY;
      el nz, nzz;
      A::lds_get(nz, SLOT_Z);
      A::lds_get(nzz, SLOT_ZZ);
      sel(sZ, nz, even[b]);
      sel(sZZ, nzz, even[b]);
      A::lds_put(SLOT_Z, sZ);
      A::lds_put(SLOT_ZZ, sZZ);
    }
    // to affine
    el Zf, zinv, zz, t3, ax, ay;
    A::lds_get(Zf, SLOT_Z);
    bad |= is0(Zf);
    if (!valid | bad) return false;
    inv(zinv, Zf);
    A::sqr(zz, zinv);
    A::mul(ax, V.X, zz);
    A::mul(t3, zz, zinv);
    A::mul(ay, V.Y, t3);
    fp<N> x, y;
    A::to_words(x, ax);
    A::to_words(y, ay);
    fp_store_be<N>(out, x);
    fp_store_be<N>(out + NB, y);
    return true;
  }

  // ---- element_from_hash (curve_from_hash, ecc/curve.c:455-482) -----------------------------------------------------
  // The routine of group_ops.cuh (g_from_hash_lane: digest expansion, the wave-cooperative search for the first x of the
  // chain with a point above it, the canonical sign) with its two long computations in limb form: the power t^((q+1)/4)
  // of a square-root attempt (510 squarings + 255 products for a.param) and the cofactor multiplication [h] (x, y) -- the
  // windowed ladder above with h's bytes as the scalar (353 bits for a.param) instead of a bit-by-bit double-and-add.
  struct LimbSqrt {
    static PBC_DEV void attempt(fp<N> &y, bool &ok, const fp<N> &t) {
      el a, r = one_el();
      A::to_el(a, t);
      for (int i = c_curve.sqrt_bits - 1; i >= 0; i--) {
        A::sqr(r, r);                  // P-class in, P-class out
        if ((c_curve.sqrt_e[i >> 5] >> (i & 31)) & 1) A::mul(r, r, a);
      }
      A::to_words(y, r);
      fp<N> yy;
      fp_sqr<N>(yy, y);
      ok = fp_eq<N>(yy, t);
    }
  };
  // false (nothing written): the cofactor ladder met an exceptional point -- the caller runs g_from_hash_lane for the lane
  static PBC_DEV bool from_hash_lane(uint8_t *out, const uint8_t *data, int hlen) {
    constexpr int NB = 4 * N;
    fp<N> ca, cb, x, fx, fy;
    fp_set<N>(ca, c_curve.a);
    fp_set<N>(cb, c_curve.b);
    fq_from_hash_lane<N>(x, data, hlen);
    g_hash_search<N, LimbSqrt>(fx, fy, x, ca, cb);
    {                                  // canonical y odd
      fp<N> o, c, ny;
#pragma unroll
      for (int i = 0; i < N; i++) o.v[i] = (i == 0);
      fp_mul<N>(c, fy, o);
      fp_neg<N>(ny, fy);
      fp_cmov<N>(fy, ny, ((c.v[0] & 1) == 0) & !fp_is0<N>(fy));
    }
    // [h] (fx, fy): the cofactor as a big-endian scalar record of whole words
    __attribute__((aligned(16))) uint8_t pt[2 * NB], zb[4 * N + 8];     // (h = (q + 1) / r has fewer bytes than q)
    fp_store_be<N>(pt, fx);
    fp_store_be<N>(pt + NB, fy);
    const int zlen = (c_curve.cofbits + 7) >> 3;           // (no padding: every byte is two more windows of the ladder)
    for (int i = 0; i < zlen; i++) zb[zlen - 1 - i] = (uint8_t) (c_curve.cofac[i >> 2] >> (8 * (i & 3)));
    return gmul_lane(out, pt, zb, zlen);
  }

  // out = a^k in GT for a of norm 1; returns false (nothing written) for any other element
  static PBC_DEV bool gt_pow_lane(uint8_t *out, const uint8_t *a, const uint8_t *z, int zlen) {
    constexpr int NB = 4 * N;
    el ax, ay, P, two, v0, v1;
    bool unitary;
    {
      fp<N> x, y, one, t0, t1;
      fp_load_be<N>(x, a);
      fp_load_be<N>(y, a + NB);
      fp_set<N>(one, fpk<N>().one);
      fp_sqr<N>(t0, x);
      fp_sqr<N>(t1, y);
      fp_add<N>(t0, t0, t1);
      unitary = fp_eq<N>(t0, one);
      A::to_el(ax, x);
      A::to_el(ay, y);
      fp_dbl<N>(one, one);
      A::to_el(two, one);              // canonical
    }
    A::template shl<1>(P, ax);
    A::norm(P, P);                     // B 2, almost normalised
    v0 = two;
    v1 = P;
    uint32_t kw[KW];
    load_scalar(kw, z, zlen);
    for (int j = 8 * zlen - 1; j >= 0; j--) {
      const bool bit = scalar_bits(kw, j, 1u) != 0;
      el m, s;
      if ((j & 15) == 0) pbc_fair_tick<PBC_A_FAIR_BIT>();
      A::mul(m, v0, v1);
      A::subk(m, m, P, K4);            // P almost normalised, B 2.1
      A::norm(m, m);                   // V_(2n+1) = V_n V_(n+1) - P: B 5.1
      s = v0;
      sel(s, v1, bit);
      A::sqr(s, s);
      A::subk(s, s, two, K2);
      A::norm(s, s);                   // V_(2n) or V_(2n+2): B 3.5
      v0 = s;
      sel(v0, m, bit);                 // bit: (V_(2n+1), V_(2n+2)), else (V_(2n), V_(2n+1))
      v1 = m;
      sel(v1, s, bit);
    }
    // Re a^k = V_k / 2,  Im a^k = -(2 V_(k+1) - P V_k) / (4 Im a)     (P^2 - 4 = -4 (Im a)^2 for norm 1; Im a = 0: a = +-1)
    el t, w, yi;
    A::mul(t, v0, P);
    A::template shl<1>(v1, v1);        // u 2, B 10.2
    A::subk(v1, v1, t, K2);
    A::norm(v1, v1);                   // B 12.2
    inv(yi, ay);
    A::mul(w, v1, yi);
    fp<N> x, y;
    A::to_words(y, w);
    fp_halve<N>(y, y);
    fp_halve<N>(y, y);
    fp_neg<N>(y, y);
    A::to_fp(x, v0);
    fp_halve<N>(x, x);
    if (unitary) {
      fp_store_be<N>(out, x);
      fp_store_be<N>(out + NB, y);
    }
    return unitary;
  }
};

}  // namespace pbc
