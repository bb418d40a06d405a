// pairing_aw.cuh -- element_pairing, type a (512-bit q), ONE PAIRING PER WAVEFRONT: the low-latency path for small
// batches (include/pbc_hip.h "batch sizes").  The throughput kernels (pairing_al.cuh) give every lane a whole pairing:
// 2.1 M dependent multiply-adds, 6.4 ms however small the batch.  Here an F_q element is ONE register: lane j of the
// wave holds limb j (29 bits, 18 lanes in use), sums and differences are single instructions, and the Montgomery
// product runs across the lanes:
//     for i in 0..17:   acc_j += a_i b_j          a_i: v_readlane -> SGPR operand of v_mad_u64_u32
//                       m = acc_0 (-1/q) mod 2^29  v_readfirstlane
//                       acc_j += m q_j
//                       acc_j = (acc_j >> 29) + (acc_{j+1} mod 2^29)      one DPP wave shift
// Lane j holds column i + j in step i: its own carry is already where it belongs after the shift, only the low 29
// bits travel, and what a lane keeps between steps fits 32 bits.  After 18 steps lane j holds limb j of
// a b / R + (multiple of q); carry passes (DPP, wave-uniform loop on a ballot) bring the limbs below 2^29.  Seven
// vector instructions per step; about 1 300 cycles per product against 5 900 for the 648 multiply-adds of a lane-local
// product at one wave per SIMD.
// The ALGORITHM is AL<N>'s, operation for operation (double_step, add_step, final_exp: same sums, same borrowed
// constants, same normalisations, hence the same bounds -- in the host mirror an element IS an AL<N>::el and carries
// AL's worst-case tracker through these routines), so the two paths agree bit for bit on every input, the invalid
// ones included.  Only the inversion differs: Fermat's x^(q-2) with 4-bit windows on the wave's products.
// Boundaries (bytes <-> limbs, validity, the final halvings) run the word-form routines on lane 0 through LDS.
#pragma once
#include "pairing_al.cuh"

namespace pbc {

constexpr int AW_SLOTS = 24;                 // LDS words per wave: AW_SLOTS x 32 (4 inputs / 2 outputs, 15 table entries)
#ifndef PBC_HOSTSIM
template <int N> __shared__ uint32_t g_lds_aw_t[AW_SLOTS * 32];
#define g_lds_aw g_lds_aw_t<N>
#endif

template <int N>
struct AW {
  typedef AL<N> A;
  typedef typename A::el el;
  static constexpr int L = A::L;
  static constexpr uint32_t MASK = A::MASK;
  enum { K2 = A::K2, K4 = A::K4, K8 = A::K8, K12 = A::K12, K16 = A::K16 };

#ifdef PBC_HOSTSIM
  // ---- host mirror: an element is all L limbs (AL's el with its tracker); the lanes run in a loop ------------------
  typedef el W;
  void init() {}
  static W add(const W &a, const W &b) { W r; A::add(r, a, b); return r; }
  template <int S> static W shl(const W &a) { W r; A::template shl<S>(r, a); return r; }
  static W subk(const W &a, const W &b, int k) { W r; A::subk(r, a, b, k); return r; }
  static W negk(const W &b, int k) { W r; A::negk(r, b, k); return r; }
  static W norm(const W &a) { W r; A::norm(r, a); return r; }
  // the lane recurrence of the device product, lanes 0 .. L (lane L: the zero lane above the top limb)
  static void lanes_sop(W &r, const W *a, const W *b, int terms) {
    const FpK<N> &K = fpk<N>();
    uint64_t acc[L + 1] = {0};
    PBC_COUNT_MACS((terms + 1) * L * L);
    for (int i = 0; i < L; i++) {
      for (int t = 0; t < terms; t++)
        for (int j = 0; j < L; j++) {
          const uint64_t p = (uint64_t) a[t].l[i] * b[t].l[j];
          if (acc[j] + p < p) A::hs_fail("wave product: accumulator overflow", (double) i);
          acc[j] += p;
        }
      const uint32_t m = ((uint32_t) acc[0] * K.ninv29) & MASK;
      for (int j = 0; j < L; j++) acc[j] += (uint64_t) m * K.p29[j];
      uint32_t lo[L + 1];
      for (int j = 0; j <= L; j++) lo[j] = (uint32_t) acc[j] & MASK;
      if (lo[0]) A::hs_fail("wave product: column not cleared", (double) i);
      for (int j = 0; j < L; j++) {
        if (acc[j] >> 60) A::hs_fail("wave product: accumulator above 2^60", (double) i);
        acc[j] = (acc[j] >> 29) + lo[j + 1];                   // 32 bits on the device
        if (acc[j] >> 32) A::hs_fail("wave product: carried limb above 32 bits", (double) j);
      }
    }
    for (int j = 0; j < L; j++) {
      if (acc[j] >> 32) A::hs_fail("wave product: limb above 32 bits", (double) j);
      r.l[j] = (uint32_t) acc[j];
    }
    strict_limbs(r);
  }
  static void strict_limbs(W &r) {            // carry passes until every limb below the top is under 2^29
    for (;;) {
      bool any = false;
      uint32_t c = 0;
      for (int j = 0; j < L; j++) {
        const uint32_t t = r.l[j];
        r.l[j] = (j < L - 1 ? (t & MASK) : t) + c;
        c = j < L - 1 ? t >> 29 : 0;
      }
      for (int j = 0; j < L - 1; j++) any |= r.l[j] > MASK;
      if (!any) break;
    }
  }
  static W mul(const W &a, const W &b) {
    W r;
    A::hs_limbs(a); A::hs_limbs(b); A::hs_cols(a.hs_u * b.hs_u);
    const double B = 1 + a.hs_B * b.hs_B / 1024;
    lanes_sop(r, &a, &b, 1);
    A::hs_set(r, A::U_STRICT, B);
    return r;
  }
  static W sqr(const W &a) { return mul(a, a); }
  static W sop2(const W &a0, const W &b0, const W &a1, const W &b1) {
    W r;
    const W x[2] = {a0, a1}, y[2] = {b0, b1};
    A::hs_limbs(a0); A::hs_limbs(b0); A::hs_limbs(a1); A::hs_limbs(b1); A::hs_cols(a0.hs_u * b0.hs_u + a1.hs_u * b1.hs_u);
    const double B = 1 + (a0.hs_B * b0.hs_B + a1.hs_B * b1.hs_B) / 1024;
    lanes_sop(r, x, y, 2);
    A::hs_set(r, A::U_STRICT, B);
    return r;
  }
  static void mul2(W &r0, W &r1, const W &a, const W &b, const W &c, const W &d) { const W x = mul(a, b), y = mul(c, d); r0 = x; r1 = y; }
  static void sop2x2(W &r0, W &r1, const W &a0, const W &b0, const W &a1, const W &b1, const W &c0, const W &d0, const W &c1, const W &d1) {
    const W x = sop2(a0, b0, a1, b1), y = sop2(c0, d0, c1, d1);
    r0 = x; r1 = y;
  }
  static W uniform(const uint32_t *limbs) { W r; for (int j = 0; j < L; j++) r.l[j] = limbs[j]; A::hs_set(r, A::U_STRICT, 1.0); return r; }
  // x >= q ?  and  x - q  (x: strict limbs, value < 2q)
  static bool geq_q(const W &x) {
    const FpK<N> &K = fpk<N>();
    for (int j = L - 1; j >= 0; j--) if (x.l[j] != K.p29[j]) return x.l[j] > K.p29[j];
    return true;
  }
  static W sub_q(const W &x) {
    const FpK<N> &K = fpk<N>();
    W r;
    for (int j = 0; j < L; j++) r.l[j] = x.l[j] - K.p29[j] + (j < L - 1 ? (1u << 29) : 0) - (j > 0 ? 1 : 0);
    strict_limbs(r);
    A::hs_set(r, A::U_STRICT, 1.0);
    return r;
  }
  static bool is_zero(const W &x) { for (int j = 0; j < L; j++) if (x.l[j]) return false; return true; }
  static W tab_[16];
  static void tab_put(int e, const W &x) { tab_[e] = x; }
  static W tab_get(int e) { return tab_[e]; }
  static void to_lane0(el &r, const W &x, int) { r = x; }
  static W from_lane0(const el &x, int) { return x; }
  static bool lane0() { return true; }
  static void sync() {}
  static bool share(bool v) { return v; }
#else
  // ---- device: one limb per lane ----------------------------------------------------------------------------------
  typedef uint32_t W;
  W kk[5], qq;                                                 // this lane's limb of the five borrowed constants and of q
  PBC_DEV void init() {
    const int j = lane();
#pragma unroll
    for (int k = 0; k < 5; k++) kk[k] = j < L ? c_a.ksub[k][j] : 0u;
    qq = j < L ? fpk<N>().p29[j] : 0u;
  }
  static PBC_DEV int lane() { return (int) (threadIdx.x & 63); }
  static PBC_DEV W from_above(W x) { return (W) __builtin_amdgcn_update_dpp(0, (int) x, 0x130, 0xf, 0xf, true); }   // wave_shl:1 : lane j <- lane j + 1
  static PBC_DEV W from_below(W x) { return (W) __builtin_amdgcn_update_dpp(0, (int) x, 0x138, 0xf, 0xf, true); }   // wave_shr:1 : lane j <- lane j - 1
  static PBC_DEV W add(W a, W b) { return a + b; }
  template <int S> static PBC_DEV W shl(W a) { return a << S; }
  PBC_DEV W subk(W a, W b, int k) const { return a - b + kk[k]; }
  PBC_DEV W negk(W b, int k) const { return kk[k] - b; }
  static PBC_DEV W norm(W a) {
    const int j = lane();
    const W c = j < L - 1 ? a >> 29 : 0u;
    return (j < L - 1 ? (a & MASK) : a) + from_below(c);
  }
  static PBC_DEV W strict_limbs(W x) {
    const int j = lane();
    for (;;) {
      x = norm(x);
      if (__ballot(j < L - 1 && x > MASK) == 0) break;
    }
    return x;
  }
  // One step of the recurrence for accumulator `acc` (see the head of the file).  Accumulators stay below 2^61 (column
  // sums <= 2.55 2^58 + m q: AL's hs_cols), so acc >> 29 is one v_alignbit.  (Forming m on the scalar unit from acc_0's
  // low word and b_0 (-1/q) -- so that the two multiply-adds issue back to back -- measured 9 % SLOWER: 2.75 against
  // 2.52 ms per pairing; the scalar multiplies sit on the same dependency chain.)
  template <int TERMS>
  static PBC_DEV void lanes_step(uint32_t &nxt, uint32_t x0, W b0, uint32_t x1, W b1, W q, uint32_t ninv) {
    uint64_t acc = (uint64_t) x0 * b0 + nxt;
    if (TERMS == 2) acc += (uint64_t) x1 * b1;
    const uint32_t m = ((uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) acc) * ninv) & MASK;   // scalar unit
    acc += (uint64_t) m * q;
    const uint32_t c = __builtin_amdgcn_alignbit((uint32_t) (acc >> 32), (uint32_t) acc, 29);   // < 2^31: acc < 2^60
    nxt = c + from_above((uint32_t) acc & MASK);
  }
  template <int TERMS>
  static PBC_DEV W lanes_sop(W a0, W b0, W a1, W b1, W q) {
    const uint32_t ninv = fpk<N>().ninv29;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < L; i++)
      lanes_step<TERMS>(acc, (uint32_t) __builtin_amdgcn_readlane((int) a0, i), b0, TERMS == 2 ? (uint32_t) __builtin_amdgcn_readlane((int) a1, i) : 0u, b1, q, ninv);
    return strict_limbs(acc);
  }
  // two independent sums of products in one instruction stream: the dependency chain of a step (multiply-add, m,
  // multiply-add, shift: eight instructions of ~9 cycles each at one wave per SIMD) leaves the pipe half empty
  struct W2 { W r0, r1; };
  template <int TERMS>
  static PBC_DEV W2 lanes_sop_x2(W a0, W b0, W a1, W b1, W c0, W d0, W c1, W d1, W q) {
    const uint32_t ninv = fpk<N>().ninv29;
    uint32_t acc = 0, bcc = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      lanes_step<TERMS>(acc, (uint32_t) __builtin_amdgcn_readlane((int) a0, i), b0, TERMS == 2 ? (uint32_t) __builtin_amdgcn_readlane((int) a1, i) : 0u, b1, q, ninv);
      lanes_step<TERMS>(bcc, (uint32_t) __builtin_amdgcn_readlane((int) c0, i), d0, TERMS == 2 ? (uint32_t) __builtin_amdgcn_readlane((int) c1, i) : 0u, d1, q, ninv);
    }
    W x = acc, y = bcc;
    const int j = lane();
    for (;;) {
      x = norm(x);
      y = norm(y);
      if (__ballot(j < L - 1 && (x > MASK || y > MASK)) == 0) break;
    }
    return W2{x, y};
  }
  static __device__ __noinline__ W2 mul2_fn(W a, W b, W c, W d, W q) { return lanes_sop_x2<1>(a, b, 0, 0, c, d, 0, 0, q); }
  static __device__ __noinline__ W2 sop2x2_fn(W a0, W b0, W a1, W b1, W c0, W d0, W c1, W d1, W q) { return lanes_sop_x2<2>(a0, b0, a1, b1, c0, d0, c1, d1, q); }
  PBC_DEV void mul2(W &r0, W &r1, W a, W b, W c, W d) const { const W2 t = mul2_fn(a, b, c, d, qq); r0 = t.r0; r1 = t.r1; }
  PBC_DEV void sop2x2(W &r0, W &r1, W a0, W b0, W a1, W b1, W c0, W d0, W c1, W d1) const { const W2 t = sop2x2_fn(a0, b0, a1, b1, c0, d0, c1, d1, qq); r0 = t.r0; r1 = t.r1; }
  static __device__ __noinline__ W mul_fn(W a, W b, W q) { return lanes_sop<1>(a, b, 0, 0, q); }
  static __device__ __noinline__ W sop2_fn(W a0, W b0, W a1, W b1, W q) { return lanes_sop<2>(a0, b0, a1, b1, q); }
  PBC_DEV W mul(W a, W b) const { return mul_fn(a, b, qq); }
  PBC_DEV W sqr(W a) const { return mul_fn(a, a, qq); }
  PBC_DEV W sop2(W a0, W b0, W a1, W b1) const { return sop2_fn(a0, b0, a1, b1, qq); }
  static PBC_DEV W uniform(const uint32_t *limbs) { const int j = lane(); return j < L ? limbs[j] : 0u; }
  PBC_DEV bool geq_q(W x) const {
    const uint32_t q = qq;
    const uint64_t gt = __ballot(x > q), lt = __ballot(x < q);
    return gt >= lt;                                           // the highest differing limb decides; equal: x = q
  }
  PBC_DEV W sub_q(W x) const {
    const int j = lane();
    W r = x - qq + (j < L - 1 ? (1u << 29) : 0u) - (j > 0 && j < L ? 1u : 0u);
    return strict_limbs(r);
  }
  static PBC_DEV bool is_zero(W x) { return __ballot(x != 0) == 0; }
  static PBC_DEV void tab_put(int e, W x) { if (lane() < 32) g_lds_aw[(6 + e) * 32 + lane()] = x; }
  static PBC_DEV W tab_get(int e) { return lane() < 32 ? g_lds_aw[(6 + e) * 32 + lane()] : 0u; }
  static PBC_DEV void sync() { __syncthreads(); }
  static PBC_DEV bool lane0() { return lane() == 0; }
  // every lane's limb -> LDS slot; lane 0 (after sync) reads the whole element
  static PBC_DEV void put_slot(W x, int slot) { if (lane() < 32) g_lds_aw[slot * 32 + lane()] = x; }
  static PBC_DEV void slot_to_el(el &r, int slot) {
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = g_lds_aw[slot * 32 + i];
  }
  static PBC_DEV void el_to_slot(const el &x, int slot) {
#pragma unroll
    for (int i = 0; i < L; i++) g_lds_aw[slot * 32 + i] = x.l[i];
#pragma unroll
    for (int i = L; i < 32; i++) g_lds_aw[slot * 32 + i] = 0;
  }
  static PBC_DEV W get_slot(int slot) { return lane() < 32 ? g_lds_aw[slot * 32 + lane()] : 0u; }
  static PBC_DEV bool share(bool v) { return __builtin_amdgcn_readfirstlane((int) v) != 0; }
#endif

  // value < 2q, any limbs the product accepts -> the canonical residue, strict limbs (what to_fp + to_el give AL)
  PBC_DEV W canon(const W &x, const W &oneR) {
    W t = mul(x, oneR);
    if (geq_q(t)) t = sub_q(t);
    return t;
  }
  // x^(q-2), x canonical and non-zero (a zero stays zero): 4-bit windows, the table in LDS
  PBC_DEV W invert(const W &x, const W &oneR) {
    const FpK<N> &K = fpk<N>();
    W t = x;
    tab_put(1, x);
    for (int e = 2; e < 16; e++) {
      t = mul(t, x);
      tab_put(e, t);
    }
    sync();
    W r = oneR;
    const int top = (int) K.pbits - 1;
    for (int w = top / 4; w >= 0; w--) {
      if (w != top / 4) {
        r = sqr(r); r = sqr(r); r = sqr(r); r = sqr(r);
      }
      const uint32_t d = (K.pm2[w >> 3] >> (4 * (w & 7))) & 15;
      if (d) r = mul(r, tab_get((int) d));
    }
    return canon(r, oneR);
  }

  struct state {
    W fx, fy, X, Y, Z, ZZ, Qx, Qy;
  };
  PBC_DEV void fsqr(state &s) {                              // AL::fsqr
    const W e0 = add(s.fx, s.fy);
    const W e1 = norm(subk(s.fx, s.fy, K2));
    const W d = shl<1>(s.fx);
    mul2(s.fx, s.fy, e0, e1, d, s.fy);
  }
  PBC_DEV void fmul(state &s, const W &lx, const W &ly) {    // AL::fmul: two lazy sums of two products
    const W nfy = norm(negk(s.fy, K2));
    sop2x2(s.fx, s.fy, s.fx, lx, nfy, ly, s.fx, ly, s.fy, lx);
  }
  // AL::double_step, its independent products issued two at a time
  PBC_DEV void double_step(state &s) {
    W XX, Z4, YY, t0, t1, lx, ly, Z3, S1, MM, Y4;
    fsqr(s);
    mul2(XX, Z4, s.X, s.X, s.ZZ, s.ZZ);
    W M = add(add(shl<1>(XX), XX), Z4);
    M = norm(M);
    mul2(YY, t0, s.Y, s.Y, s.Qx, s.ZZ);
    t0 = add(t0, s.X);
    mul2(lx, Z3, M, t0, shl<1>(s.Y), s.Z);
    t1 = shl<1>(YY);
    lx = norm(subk(lx, t1, K4));
    mul2(t1, S1, Z3, s.ZZ, YY, shl<1>(s.X));
    s.Z = Z3;
    mul2(s.ZZ, MM, Z3, Z3, M, M);
    mul2(ly, Y4, t1, s.Qy, YY, YY);
    fmul(s, lx, ly);
    t1 = shl<2>(S1);
    s.X = norm(subk(MM, t1, K8));
    t1 = shl<1>(S1);
    const W Wd = norm(subk(t1, s.X, K12));
    t0 = mul(M, Wd);
    t1 = norm(shl<3>(Y4));
    s.Y = norm(subk(t0, t1, K12));
  }
  PBC_DEV void add_step(state &s, const W &x2, const W &y2) {   // AL::add_step
    W t0 = mul(x2, s.ZZ);
    const W H = norm(subk(t0, s.X, K16));
    t0 = mul(s.Z, s.ZZ);
    t0 = mul(y2, t0);
    const W R = norm(subk(t0, s.Y, K16));
    W Z3 = mul(H, s.Z);
    t0 = add(s.Qx, x2);
    W lx = mul(R, t0);
    t0 = mul(Z3, y2);
    lx = norm(subk(lx, t0, K2));
    const W ly = mul(Z3, s.Qy);
    W HH = sqr(H);
    W HHH = mul(HH, H);
    t0 = mul(s.X, HH);
    W t1 = sqr(R);
    t1 = norm(subk(t1, HHH, K2));
    HH = shl<1>(t0);
    t1 = norm(subk(t1, HH, K4));
    t0 = norm(subk(t0, t1, K16));
    t0 = mul(R, t0);
    HHH = mul(s.Y, HHH);
    s.Y = norm(subk(t0, HHH, K2));
    s.X = t1;
    s.Z = Z3;
    s.ZZ = sqr(Z3);
    fmul(s, lx, ly);
  }

  // f^((q^2-1)/r): AL::final_exp with the inversion on the wave.  Leaves (v0 R-scaled, y) for the word-form tail:
  //     out.x = v0 / 2,   out.y = -y / 4
  PBC_DEV void final_exp(W &ox, W &oy, const state &s, const W &oneR) {
    const W a2 = sqr(s.fx), b2 = sqr(s.fy);
    const W Nn = norm(add(a2, b2));
    const W Av = norm(subk(a2, b2, K2));
    W B = mul(shl<1>(s.fx), s.fy);
    B = norm(negk(B, K2));
    B = canon(B, oneR);
    if (is_zero(B)) B = oneR;                                  // f^(q-1) = +-1: invert N * 1 instead (a_final_exp)
    W t = canon(mul(Nn, B), oneR);
    t = invert(t, oneR);                                       // 1/(N B), canonical
    W w = mul(t, B);                                           // 1/N
    const W g0 = mul(Av, w);
    t = mul(t, Nn);                                            // 1/B
    w = mul(t, Nn);                                            // N/B
    const W P = norm(shl<1>(g0));
    W two = strict2(add(oneR, oneR));
    if (geq_q(two)) two = sub_q(two);
    W v0 = two, v1 = P;
    for (int j = c_a.hbits - 1; j >= 0; j--) {
      const bool bit = j ? ((c_a.h[j >> 5] >> (j & 31)) & 1) : false;
      const W v = bit ? v1 : v0;
      W m, sq;
      mul2(m, sq, v0, v1, v, v);
      m = norm(subk(m, P, K4));
      sq = norm(subk(sq, two, K2));
      if (bit) {
        v1 = sq;
        v0 = m;
      } else {
        v0 = sq;
        v1 = m;
      }
    }
    t = mul(v0, P);
    v1 = shl<1>(v1);
    v1 = norm(subk(v1, t, K2));
    oy = mul(v1, w);
    ox = mul(v0, oneR);
  }
  // 2 R mod q from R + R: strict limbs of a value < 2q (the sum's limbs are below 2^30)
#ifdef PBC_HOSTSIM
  static W strict2(const W &x) { W r = x; strict_limbs(r); A::hs_set(r, A::U_STRICT, 2.0); return r; }
#else
  static PBC_DEV W strict2(W x) { return strict_limbs(x); }
#endif

  // element_pairing, one wave: gt <- e(g1, g2)
  PBC_DEV void pairing_wave(uint8_t *gt, const uint8_t *g1, const uint8_t *g2) {
    constexpr int NB = 4 * N;
    bool valid = false;
    state s;
    init();
    W Px, Py, oneR;
    {
      el e[5];
      if (lane0()) {
        fp<N> px, py, qx, qy, one;
        fp_load_be<N>(px, g1);
        fp_load_be<N>(py, g1 + NB);
        fp_load_be<N>(qx, g2);
        fp_load_be<N>(qy, g2 + NB);
        valid = a_first_arg_ok<N>(px, py) & a_on_curve<N>(qx, qy);
        fp_set<N>(one, fpk<N>().one);
        A::to_el(e[0], px);
        A::to_el(e[1], py);
        A::to_el(e[2], qx);
        A::to_el(e[3], qy);
        A::to_el(e[4], one);
#ifndef PBC_HOSTSIM
        for (int i = 0; i < 5; i++) el_to_slot(e[i], i);
#endif
      }
#ifdef PBC_HOSTSIM
      Px = e[0]; Py = e[1]; s.Qx = e[2]; s.Qy = e[3]; oneR = e[4];
#else
      sync();
      Px = get_slot(0); Py = get_slot(1); s.Qx = get_slot(2); s.Qy = get_slot(3); oneR = get_slot(4);
      sync();
#endif
    }
    s.X = Px; s.Y = Py; s.Z = oneR; s.ZZ = oneR; s.fx = oneR;
#ifdef PBC_HOSTSIM
    { el z; for (int i = 0; i < L; i++) z.l[i] = 0; A::hs_set(z, A::U_STRICT, 1.0); s.fy = z; }
#else
    s.fy = 0;
#endif
    for (int i = c_a.exp2 - 1; i >= 0; i--) {
      double_step(s);
      if (i == c_a.exp1) {
        W y2 = Py;
        if (c_a.sign1 < 0) y2 = norm(negk(y2, K2));
        add_step(s, Px, y2);
      }
    }
    W ox, oy;
    final_exp(ox, oy, s, oneR);
#ifndef PBC_HOSTSIM
    put_slot(ox, 0);
    put_slot(oy, 1);
    sync();
#endif
    if (lane0()) {
      el ex, ey;
#ifdef PBC_HOSTSIM
      ex = ox; ey = oy;
#else
      slot_to_el(ex, 0);
      slot_to_el(ey, 1);
#endif
      fp2<N> out;
      fp<N> x, y;
      A::to_words(y, ey);
      fp_halve<N>(y, y);
      fp_halve<N>(y, y);
      fp_neg<N>(out.y, y);
      A::to_words(x, ex);
      fp_halve<N>(out.x, x);
      a_store_gt<N>(gt, out, valid);
    }
  }
};
#ifdef PBC_HOSTSIM
template <int N> typename AW<N>::W AW<N>::tab_[16];
#endif

}  // namespace pbc
