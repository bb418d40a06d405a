// group_l5.cuh -- element_mul_zn on G1 of the 5-word fields (d159.param, f.param) in LIMB FORM: the regular signed-window
// ladder of group_ops.cuh (ec_mul_win_lane) on six 29-bit limbs per element, in the redundant representation of the
// type d / type f Miller steps (pairing_d.cuh d_dbl_core_l / d_add_core_l, pairing_f.cuh pl_*): additions without
// carries, a - b as a + K - b with K = c q in borrowed form, a parallel carry pass where the next product needs it, one
// conversion to words per inversion and at the store.  The word-form policy (FqOps<5>) spends 55 multiply-adds per
// product inside ~230 instructions of conversions, carries and conditional subtractions; here a product is 72
// multiply-adds in ~110 instructions.
// The borrowed constants are those of the pairing kernels, found in the object's constant block by KP (type d: behind
// the x-power table of DConst; type f: FConst::pk29); the same host check that admits a q for those kernels admits it
// here (DConst::limb_ok / FConst::pl_ok).  Bounds: per line below (u = limb size in units of 2^29, B = value in units
// of q), asserted in the host mirror by the same worst-case tracker as in those files.
#pragma once
#include "group_ops.cuh"

namespace pbc {

struct KPd { static PBC_DEV uint32_t kc(int idx, int l) { return c_d.xpwr29[64 + idx * 6 + l]; } };   // TypeMNT<5, 3>::KSUB_OFF
struct KPf { static PBC_DEV uint32_t kc(int idx, int l) { return c_f.pk29[idx][l]; } };

template <int N, class KP>
struct GL {
  static constexpr int FL = Limbs29<N>::L;
  static_assert(N == 5 && FL == 6, "written for the 5-word fields");
  typedef fl<N> el;
  enum { K2 = 0, K4 = 1, K16 = 2, K32 = 3 };                  // (c, D) = (2, 1), (4, 2), (16, 2), (32, 2)
#ifdef PBC_HOSTSIM
  static constexpr double U_STRICT = 1.0 - 1.0 / 536870912.0, U_ALMOST = 1.0 + 7.0 / 536870912.0;
  static constexpr double KC[4] = {2, 4, 16, 32}, KD[4] = {1, 2, 2, 2};
  static constexpr double SLACK = 16384.0;                    // R / q > 2^14
  static void hs_fail(const char *what, double v) { fprintf(stderr, "hostsim: limb-form G1 ladder (5-word fields): %s (%g)\n", what, v); abort(); }
  static void hs_limbs(const el &a) {
    for (int i = 0; i < FL; i++)
      if ((double) a.l[i] > a.hs_u * 536870912.0) hs_fail("limb above its tracked bound", a.hs_u);
  }
  static void hs_set(el &r, double u, double B) { r.hs_u = u; r.hs_B = B; hs_limbs(r); }
  static void hs_dom(const el &b, int k) {
    hs_limbs(b);
    if (b.hs_u > KD[k] * U_STRICT + 1e-12) hs_fail("subtrahend limbs not dominated", b.hs_u);
    if ((KC[k] - b.hs_B) * 128.0 < KD[k] + 1) hs_fail("subtrahend value not dominated", b.hs_B);   // q >= 2^152: limb_ok / pl_ok
  }
  static void hs_cols(double s) { if (s > (64.0 - FL) / FL - 0.01) hs_fail("column capacity", s); }
#define GL_HS(...) __VA_ARGS__
#else
#define GL_HS(...)
#endif
  static PBC_DEV el kconst(int idx) {
    el r;
#pragma unroll
    for (int l = 0; l < FL; l++) r.l[l] = KP::kc(idx, l);
    return r;
  }
  static PBC_DEV void from_fq(el &r, const fp<N> &a) { to_limbs<N>(r, a); GL_HS(hs_set(r, U_STRICT, 1.0);) }
  static PBC_DEV void add(el &r, const el &a, const el &b) {
#pragma unroll
    for (int i = 0; i < FL; i++) r.l[i] = a.l[i] + b.l[i];
    GL_HS(if (a.hs_u + b.hs_u >= 8) hs_fail("sum overflows 32 bits", a.hs_u + b.hs_u); hs_set(r, a.hs_u + b.hs_u, a.hs_B + b.hs_B);)
  }
  template <int S>
  static PBC_DEV void shl(el &r, const el &a) {
#pragma unroll
    for (int i = 0; i < FL; i++) r.l[i] = a.l[i] << S;
    GL_HS(if (a.hs_u * (1 << S) >= 8) hs_fail("shift overflows 32 bits", a.hs_u); hs_set(r, a.hs_u * (1 << S), a.hs_B * (1 << S));)
  }
  static PBC_DEV void subk(el &r, const el &a, const el &b, int k) {      // a + K_k - b
    const el K = kconst(k);
    GL_HS(hs_dom(b, k); const double u = a.hs_u + KD[k] + 1, B = a.hs_B + KC[k]; if (u >= 8) hs_fail("difference overflows 32 bits", u);)
#pragma unroll
    for (int i = 0; i < FL; i++) r.l[i] = a.l[i] - b.l[i] + K.l[i];
    GL_HS(hs_set(r, u, B);)
  }
  static PBC_DEV void negk(el &r, const el &b, int k) {
    const el K = kconst(k);
    GL_HS(hs_dom(b, k);)
#pragma unroll
    for (int i = 0; i < FL; i++) r.l[i] = K.l[i] - b.l[i];
    GL_HS(hs_set(r, KD[k] + 1, KC[k]);)
  }
  static PBC_DEV void norm(el &r, const el &a) {              // parallel carry pass: limbs <= 2^29 + 6
    uint32_t c = 0;
    GL_HS(hs_limbs(a); const double B = a.hs_B; if (a.hs_u >= 8) hs_fail("normalising limbs above 32 bits", a.hs_u);)
#pragma unroll
    for (int i = 0; i < FL; i++) {
      const uint32_t t = a.l[i];
      r.l[i] = (i < FL - 1 ? (t & Limbs29<N>::MASK) : t) + c;
      c = t >> 29;
    }
    GL_HS(hs_set(r, U_ALMOST, B);)
  }
  // UNITS: the product of the operands' limb sizes (a column holds 9)
  template <int UNITS>
  static PBC_DEV void mul(el &r, const el &a, const el &b) {
    const el x[1] = {a}, y[1] = {b};
    GL_HS(hs_limbs(a); hs_limbs(b); hs_cols(a.hs_u * b.hs_u); if (a.hs_u * b.hs_u > UNITS + 0.001) hs_fail("more product units than declared", a.hs_u * b.hs_u);
          const double B = 1 + a.hs_B * b.hs_B / SLACK;)
    sop_limbs<N, 1, UNITS - 1>(r, x, y);
    GL_HS(hs_set(r, U_STRICT, B);)
  }
  static PBC_DEV void sqr(el &r, const el &a) {               // a (almost) normalised
    GL_HS(const el x[1] = {a}; hs_limbs(a); if (a.hs_u > U_ALMOST) hs_fail("squaring an unnormalised element", a.hs_u); hs_sop_check<N>(x, x, 1);
          const double B = 1 + a.hs_B * a.hs_B / SLACK;)
    sqr_limbs<N>(r.l, a.l);
    GL_HS(hs_set(r, U_STRICT, B);)
  }
  static PBC_DEV void sop2(el &r, const el &a0, const el &b0, const el &a1, const el &b1) {     // a0 b0 + a1 b1, one reduction
    const el x[2] = {a0, a1}, y[2] = {b0, b1};
    GL_HS(hs_limbs(a0); hs_limbs(b0); hs_limbs(a1); hs_limbs(b1); hs_cols(a0.hs_u * b0.hs_u + a1.hs_u * b1.hs_u);
          if (a0.hs_u * b0.hs_u + a1.hs_u * b1.hs_u > 2.01) hs_fail("two-term sum of unnormalised operands", a0.hs_u);
          const double B = 1 + (a0.hs_B * b0.hs_B + a1.hs_B * b1.hs_B) / SLACK;)
    sop_limbs<N, 2, 0>(r, x, y);
    GL_HS(hs_set(r, U_STRICT, B);)
  }
  // any class -> fully reduced words (one product by R mod q brings the value below 2q)
  static PBC_DEV void to_fq(fp<N> &r, const el &a) {
    el one, t;
    fp<N> o;
    fp_set<N>(o, fpk<N>().one);
    from_fq(one, o);
    mul<8>(t, a, one);
    from_limbs<N>(r, t);
  }
  static PBC_DEV void sel(el &r, const el &a, bool c) {       // r <- a where c (per lane)
#pragma unroll
    for (int i = 0; i < FL; i++) r.l[i] = c ? a.l[i] : r.l[i];
    GL_HS(r.hs_u = r.hs_u > a.hs_u ? r.hs_u : a.hs_u; r.hs_B = r.hs_B > a.hs_B ? r.hs_B : a.hs_B;)
  }
  static PBC_DEV void inv(el &r, const el &a) {
    fp<N> w;
    to_fq(w, a);
    fp_inv<N>(w, w);
    from_fq(r, w);
  }
  static PBC_DEV bool is0(const el &a) {
    fp<N> w;
    to_fq(w, a);
    return fp_is0<N>(w);
  }

  // V <- 2V on y^2 = x^3 + a x + b, Jacobian (d_dbl_core_l without the tangent's coefficients).
  // In and out: X, Y almost normalised with B <= 18, Z with u <= 2, B <= 3.
  static PBC_DEV void dbl(el &X, el &Y, el &Z, const el &ca, bool a_zero) {
    el ZZ, XX, YY, M, t0, t1, S1, Z3, W, Xn, Yn;
    sqr(XX, X);
    sqr(YY, Y);
    shl<1>(M, XX);
    add(M, M, XX);                     // 3 X^2: u 3
    if (!a_zero) {                     // wave-uniform (type f: a = 0)
      mul<4>(ZZ, Z, Z);
      sqr(t0, ZZ);
      mul<1>(t0, t0, ca);              // a Z^4
      add(M, M, t0);                   // u 4, B 5
    }
    norm(M, M);
    mul<2>(Z3, Y, Z);
    shl<1>(Z3, Z3);                    // 2YZ: u 2, B 3
    mul<1>(S1, X, YY);                 // X Y^2
    shl<3>(t1, S1);                    // 8 X Y^2: u < 8, B 12
    norm(t1, t1);
    sqr(t0, M);
    subk(t0, t0, t1, K16);             // X3 = M^2 - 2S, S = 4 X Y^2: u 4, B 17.5
    norm(Xn, t0);
    shl<2>(t1, S1);                    // S: u 4, B 6
    subk(W, t1, Xn, K32);              // S - X3: u 7, B 38
    mul<7>(t0, M, W);
    sqr(t1, YY);
    shl<3>(t1, t1);                    // 8 Y^4: u < 8, B 12
    norm(t1, t1);
    subk(t0, t0, t1, K16);             // Y3 = M (S - X3) - 8Y^4: u 4, B 17.5
    norm(Yn, t0);
    X = Xn;
    Y = Yn;
    Z = Z3;
  }
  // V <- V + (x2, y2), mixed (d_add_core_l without the chord's coefficients).  x2, y2: limbs <= 2 units together with
  // what they meet (strict or almost normalised), B <= 18.  Out: X almost normalised B <= 7.5, Y and Z strict.
  static PBC_DEV void madd(el &X, el &Y, el &Z, const el &x2, const el &y2) {
    el ZZ, H, Rn, HH, HHH, t0, t1, Z3, W, Xn, nY;
    mul<4>(ZZ, Z, Z);
    mul<2>(H, x2, ZZ);
    subk(H, H, X, K32);                // u 4, B 33.5
    norm(H, H);
    mul<2>(t0, Z, ZZ);
    mul<2>(t0, y2, t0);
    subk(Rn, Y, t0, K2);               // -R: u 4, B 19.5
    norm(Rn, Rn);
    mul<2>(Z3, Z, H);
    sqr(HH, H);
    mul<1>(HHH, HH, H);
    mul<1>(t0, X, HH);                 // X1 H^2
    sqr(t1, Rn);
    subk(t1, t1, HHH, K2);             // u 3, B 3.5
    shl<1>(W, t0);                     // u 2, B 3
    subk(t1, t1, W, K4);               // X3 = R^2 - H^3 - 2 X1 H^2: u 6, B 7.5
    norm(Xn, t1);
    subk(W, Xn, t0, K2);               // X3 - X1 H^2: u 3, B 9.5
    norm(W, W);
    negk(nY, Y, K32);                  // -Y1: u 3, B 32
    norm(nY, nY);
    sop2(t0, Rn, W, nY, HHH);          // Y3 = R (X1 H^2 - X3) - Y1 H^3 = Rn (X3 - X1 H^2) + (-Y1) H^3
    X = Xn;
    Y = t0;
    Z = Z3;
  }

  // out <- [z] in (records of G1; z: zlen big-endian bytes).  false: nothing written, the complete ladder takes the lane
  // (a result O, a point of small order among the table's multiples, an even scalar meeting -P).
  static PBC_DEV bool gmul_lane(uint8_t *out, const uint8_t *in, const uint8_t *z, int zlen) {
    constexpr int WIN = 4, TE = 8;
    const int NB = (int) fpk<N>().fbytes;
    const bool a_zero = c_curve.a_is_zero != 0;
    el tab[TE][2], zs[TE], cs[TE], ca, one;
    bool valid;
    {
      fp<N> x, y, t0, t1, a, b;
      fp_load_be<N>(x, in);
      fp_load_be<N>(y, in + NB);
      fp_set<N>(a, c_curve.a);
      fp_set<N>(b, c_curve.b);
      fp_sqr<N>(t0, x);
      fp_add<N>(t0, t0, a);
      fp_mul<N>(t0, t0, x);
      fp_add<N>(t0, t0, b);
      fp_sqr<N>(t1, y);
      valid = fp_eq<N>(t0, t1);
      from_fq(tab[0][0], x);
      from_fq(tab[0][1], y);
      from_fq(ca, a);
      fp_set<N>(t0, fpk<N>().one);
      from_fq(one, t0);
    }
    el X = tab[0][0], Y = tab[0][1], Z = one;
    dbl(X, Y, Z, ca, a_zero);          // 2P = (X2 : Y2 : Z2)
    const el X2 = X, Y2 = Y, Z2 = Z;
    {
      // phi(P) = (x Z2^2, y Z2^3) on the isomorphic curve where phi(2P) = (X2, Y2) is affine
      el zz, t;
      mul<4>(zz, Z2, Z2);
      mul<1>(X, tab[0][0], zz);
      mul<2>(t, zz, Z2);
      mul<1>(Y, tab[0][1], t);
      Z = one;
    }
    for (int j = 1; j < TE; j++) {
      madd(X, Y, Z, X2, Y2);
      tab[j][0] = X;
      tab[j][1] = Y;
      mul<2>(zs[j], Z, Z2);            // back on E: Z <- Z' Z2
      if (j == 1) cs[1] = zs[1];
      else mul<1>(cs[j], cs[j - 1], zs[j]);
    }
    bool bad = is0(cs[TE - 1]);        // some odd multiple (or 2P) is O: a point of small order
    el zi;
    inv(zi, cs[TE - 1]);
    for (int j = TE - 1; j >= 1; j--) {
      el zinv, zz, t, x, y;
      if (j > 1) {
        mul<1>(zinv, zi, cs[j - 1]);
        mul<1>(zi, zi, zs[j]);
      } else {
        zinv = zi;
      }
      sqr(zz, zinv);
      x = tab[j][0];
      y = tab[j][1];
      mul<1>(tab[j][0], x, zz);
      mul<1>(t, zz, zinv);
      mul<1>(tab[j][1], y, t);
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
      GL_HS(hs_set(x, U_STRICT, 2.0); hs_set(y, U_STRICT, 2.0);)
      el ny;
      negk(ny, y, K4);                 // u 3, B 4
      norm(ny, ny);
      GL_HS(y.hs_u = ny.hs_u; y.hs_B = ny.hs_B;)
      sel(y, ny, neg);
    };
    entry(X, Y, t - 1);                // the top digit is positive (bit n is set)
    Z = one;
    for (int i = t - 2; i >= 0; i--) {
      el x2, y2;
      for (int d = 0; d < WIN; d++) dbl(X, Y, Z, ca, a_zero);
      entry(x2, y2, i);
      madd(X, Y, Z, x2, y2);
    }
    {
      // even k: [k] P = [k + 1] P - P
      el sX = X, sY = Y, sZ = Z, px = tab[0][0], ny = tab[0][1];
      GL_HS(hs_set(px, U_STRICT, 1.0); hs_set(ny, U_STRICT, 1.0);)
      negk(ny, ny, K2);
      norm(ny, ny);
      madd(sX, sY, sZ, px, ny);
      sel(X, sX, even);
      sel(Y, sY, even);
      sel(Z, sZ, even);
    }
    bad |= is0(Z);
    el zinv, zz, t3, ax, ay;
    inv(zinv, Z);
    sqr(zz, zinv);
    mul<2>(ax, X, zz);
    mul<1>(t3, zz, zinv);
    mul<2>(ay, Y, t3);
    fp<N> x, y;
    to_fq(x, ax);
    to_fq(y, ay);
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

  // out <- [k_0] P_0 + ... + [k_{KB-1}] P_{KB-1} (element_pow2_zn / element_pow3_zn on G1) with ONE accumulator: the ladder above on
  // KB tables -- four doublings and KB mixed additions per window, where the composition of single-base ladders doubles 4 KB
  // times.  false: nothing written, the complete multi-scalar routine takes the lane (a base off the curve, an exceptional
  // addition anywhere -- equal or opposite bases meeting in the accumulator, a point of small order, a zero scalar -- or the
  // result O: all of these leave Z = 0).
  template <int KB>
  static PBC_DEV bool gmulk_lane(uint8_t *out, const uint8_t *const *in, const uint8_t *const *z, int zlen) {
    constexpr int WIN = 4, TE = 8;
    const int NB = (int) fpk<N>().fbytes;
    const bool a_zero = c_curve.a_is_zero != 0;
    el tab[KB][TE][2], ca, one;
    uint32_t kw[KB][kScalarWords];
    bool even[KB];
    bool valid = true, bad = false;
    {
      fp<N> a, t0;
      fp_set<N>(a, c_curve.a);
      from_fq(ca, a);
      fp_set<N>(t0, fpk<N>().one);
      from_fq(one, t0);
    }
    el X, Y, Z;
#pragma unroll 1
    for (int b = 0; b < KB; b++) {
      el zs[TE], cs[TE];
      {
        fp<N> x, y, t0, t1, a, cb;
        fp_load_be<N>(x, in[b]);
        fp_load_be<N>(y, in[b] + NB);
        fp_set<N>(a, c_curve.a);
        fp_set<N>(cb, c_curve.b);
        fp_sqr<N>(t0, x);
        fp_add<N>(t0, t0, a);
        fp_mul<N>(t0, t0, x);
        fp_add<N>(t0, t0, cb);
        fp_sqr<N>(t1, y);
        valid &= fp_eq<N>(t0, t1);
        from_fq(tab[b][0][0], x);
        from_fq(tab[b][0][1], y);
      }
      X = tab[b][0][0]; Y = tab[b][0][1]; Z = one;
      dbl(X, Y, Z, ca, a_zero);        // 2P = (X2 : Y2 : Z2)
      const el X2 = X, Y2 = Y, Z2 = Z;
      {
        // phi(P) = (x Z2^2, y Z2^3) on the isomorphic curve where phi(2P) = (X2, Y2) is affine
        el zz, t;
        mul<4>(zz, Z2, Z2);
        mul<1>(X, tab[b][0][0], zz);
        mul<2>(t, zz, Z2);
        mul<1>(Y, tab[b][0][1], t);
        Z = one;
      }
      for (int j = 1; j < TE; j++) {
        madd(X, Y, Z, X2, Y2);
        tab[b][j][0] = X;
        tab[b][j][1] = Y;
        mul<2>(zs[j], Z, Z2);          // back on E: Z <- Z' Z2
        if (j == 1) cs[1] = zs[1];
        else mul<1>(cs[j], cs[j - 1], zs[j]);
      }
      bad |= is0(cs[TE - 1]);          // some odd multiple (or 2P) is O: a point of small order
      el zi;
      inv(zi, cs[TE - 1]);
      for (int j = TE - 1; j >= 1; j--) {
        el zinv, zz, t, x, y;
        if (j > 1) {
          mul<1>(zinv, zi, cs[j - 1]);
          mul<1>(zi, zi, zs[j]);
        } else {
          zinv = zi;
        }
        sqr(zz, zinv);
        x = tab[b][j][0];
        y = tab[b][j][1];
        mul<1>(tab[b][j][0], x, zz);
        mul<1>(t, zz, zinv);
        mul<1>(tab[b][j][1], y, t);
      }
      scalar_load(kw[b], z[b], zlen);
      even[b] = (kw[b][0] & 1) == 0;
      kw[b][0] |= 1u;
      kw[b][(8 * zlen) >> 5] |= 1u << ((8 * zlen) & 31);
    }
    const int t = 2 * zlen;
    auto entry = [&](el &x, el &y, int b, int i, bool top) {
      const uint32_t v = scalar_bits(kw[b], 4 * i + 1, 15u);
      const bool neg = !top && v < 8;  // (the top digit is positive: bit n is set)
      const int idx = v < 8 ? 7 - (int) v : (int) v - 8;
      x = tab[b][idx][0];
      y = tab[b][idx][1];
      GL_HS(hs_set(x, U_STRICT, 2.0); hs_set(y, U_STRICT, 2.0);)
      el ny;
      negk(ny, y, K4);                 // u 3, B 4
      norm(ny, ny);
      GL_HS(y.hs_u = ny.hs_u; y.hs_B = ny.hs_B;)
      sel(y, ny, neg);
    };
    entry(X, Y, 0, t - 1, true);
    Z = one;
#pragma unroll 1
    for (int b = 1; b < KB; b++) {
      el x2, y2;
      entry(x2, y2, b, t - 1, true);
      madd(X, Y, Z, x2, y2);
    }
    for (int i = t - 2; i >= 0; i--) {
      for (int d = 0; d < WIN; d++) dbl(X, Y, Z, ca, a_zero);
#pragma unroll 1
      for (int b = 0; b < KB; b++) {
        el x2, y2;
        entry(x2, y2, b, i, false);
        madd(X, Y, Z, x2, y2);
      }
    }
#pragma unroll 1
    for (int b = 0; b < KB; b++) {
      // even k_b: the sum holds [k_b + 1] P_b; take P_b off
      el sX = X, sY = Y, sZ = Z, px = tab[b][0][0], ny = tab[b][0][1];
      GL_HS(hs_set(px, U_STRICT, 1.0); hs_set(ny, U_STRICT, 1.0);)
      negk(ny, ny, K2);
      norm(ny, ny);
      madd(sX, sY, sZ, px, ny);
      sel(X, sX, even[b]);
      sel(Y, sY, even[b]);
      sel(Z, sZ, even[b]);
    }
    bad |= is0(Z);
    if (!valid | bad) return false;
    el zinv, zz, t3, ax, ay;
    inv(zinv, Z);
    sqr(zz, zinv);
    mul<2>(ax, X, zz);
    mul<1>(t3, zz, zinv);
    mul<2>(ay, Y, t3);
    fp<N> x, y;
    to_fq(x, ax);
    to_fq(y, ay);
    fp_store_be<N>(out, x);
    fp_store_be<N>(out + NB, y);
    return true;
  }

  // element_pp_pow_zn on G1: out <- [z] B from the 8-bit-row table of element_pp_init (group_ops.cuh ec_pp_entry_lane:
  // affine Montgomery words), one mixed addition per byte of the scalar -- ec_pp_pow_lane in limb form.  false: the lane
  // needs the complete routine (two equal table points met: a doubling).
  static PBC_DEV bool pp_pow_lane(uint8_t *out, const uint32_t *__restrict__ tab, const uint8_t *z, int zlen) {
    const int NB = (int) fpk<N>().fbytes;
    el one;
    {
      fp<N> o;
      fp_set<N>(o, fpk<N>().one);
      from_fq(one, o);
    }
    el X = one, Y = one, Z = one;
    bool inf = true;                   // the accumulator is still O
    for (int row = 0; row < zlen; row++) {
      const uint32_t w = z[zlen - 1 - row];
      const size_t u = (size_t) row * kPpRowLen + (w ? w - 1 : 0);
      fp<N> wx, wy;
      fp_set<N>(wx, tab + u * 2 * N);
      fp_set<N>(wy, tab + (u * 2 + 1) * N);
      el x2, y2, sX = X, sY = Y, sZ = Z;
      from_fq(x2, wx);
      from_fq(y2, wy);
      madd(sX, sY, sZ, x2, y2);
      const bool take = w != 0, first = take & inf, acc = take & !inf;
      sel(X, sX, acc);
      sel(Y, sY, acc);
      sel(Z, sZ, acc);
      sel(X, x2, first);
      sel(Y, y2, first);
      inf &= !take;
    }
    const bool bad = !inf & is0(Z);
    el zinv, zz, t3, ax, ay;
    inv(zinv, Z);
    sqr(zz, zinv);
    mul<2>(ax, X, zz);
    mul<1>(t3, zz, zinv);
    mul<2>(ay, Y, t3);
    fp<N> x, y;
    to_fq(x, ax);
    to_fq(y, ay);
    if (inf) {                         // k = 0
#pragma unroll
      for (int k = 0; k < N; k++) { x.v[k] = 0; y.v[k] = 0; }
    }
    if (!bad) {
      fp_store_be<N>(out, x);
      fp_store_be<N>(out + NB, y);
    }
    return !bad;
  }
};

}  // namespace pbc
