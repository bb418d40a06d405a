// pairing_al.cuh -- Type A (a.param: 512-bit q, 16 words) with F_q elements kept in 29-bit LIMB form.
//
// element_pairing (a_pairing_proj + a_tateexp, ecc/a_param.c:1053-1198, :285-303) and pairing_pp_apply (:317-360) with
// the Jacobian Miller steps and the Lucas-sequence final exponentiation of pairing_a.cuh (a_double_step, a_add_step,
// a_final_exp: the word-form routines the other type a kernels use) -- but an element is L = 18 limbs of 29 bits
// from the moment it is loaded until it is stored.  The word-form kernel converts both operands of every product into
// limbs and the result back into words with a conditional subtraction: 150 of the 900 instructions of a product
// (measured by a what-if build: +14 % pairings/s without them, profiles/r02_notes.md).  The multiply-add pipe takes a
// v_mad_u64_u32 every 4.58 cycles per SIMD and any other VALU instruction ~2.5 (tools/mac_chain_probe.hip); this
// kernel's 2.15 M multiply-adds + 0.77 M other instructions per pairing run at 92 % of that pipe-time.
//
// Redundant representation.  R = 2^522 and q < 2^512, so there are ten bits of slack: a value is any 18 limbs whose
// sum is congruent to it; a Montgomery product of operands below 16 q comes out below 1.25 q ("P-class": limbs
// < 2^29 exactly, value < 1.5 q), additions are 18 plain v_add without carries, and there are no conditional
// subtractions at all until the result is stored.  Two bounds are tracked by hand through every routine below
// (and checked by assertions in the host mirror, tests/hostsim):
//   * limb size u, in units of 2^29: a column of the multiplier holds 18 operand products + 18 products m q, so
//         18 (sum over terms of u_x u_y) 2^58 + 18 2^58 < 2^64   <=>   sum u_x u_y <= 2.55
//     -- one operand of a product may be a sum of two normalised values (u = 2), squares need u <= 1.58;
//   * value bound B, in units of q: B(out) = 1 + sum B(x) B(y) / 1024.
// norm() is a PARALLEL carry pass (3 instructions per limb, no chain): limbs end up <= 2^29 + 6 ("almost normalised").
// A difference a - b is formed as a + (K - b) limb by limb with K = c q written so that every limb dominates the
// corresponding limb of b (borrowed form: limb_i(c q) + D 2^29 - D); the host precomputes the five K it needs.
//
// Calling convention of the out-of-line products.  An operand is 18 VGPRs and the ABI passes 31: the first operand
// travels in registers, the second as 12 registers + 6 words of LDS.  The Miller accumulator f and the coordinates
// Z, Z^2 of the running point live in LDS for the whole loop (four slots): 72 registers less in the point arithmetic
// (a value that lives across a call must sit in one of the ABI's 108 callee-saved VGPRs or be spilled); Q sits in the
// lane's private memory.  LDS per lane: 4 x 18 + 6 words = 312 B -> 39 KB per 128-lane workgroup, four workgroups
// (two waves per SIMD) per CU.  Three product bodies in all (product, square, two-term sum against f): the Miller loop's
// code is 31 KB; variants with fused f^2 / f * line bodies (64 KB per iteration) ran 3 % slower (profiles/r02_notes.md).
#pragma once
#include "pairing_a.cuh"

namespace pbc {

constexpr int AL_LANES = 128;
// Slices of the time-sliced priorities between the two waves of a SIMD (fp.cuh pbc_fair_tick; the kernels on these routines
// are launched with resident workgroups): 2^23 cycles = 3.5 ms.  Same-box A/B on 2^20 pairings: none (one workgroup per
// 128 units) 87.6 ms, 2^21 82.1, 2^23 81.2 - 82.0, 2^24 82.5, 2^25 83.3, 2^26 84.4.
#ifndef PBC_A_FAIR_BIT
#define PBC_A_FAIR_BIT 23
#endif
#ifndef PBC_A_QVEC
#define PBC_A_QVEC 1
#endif
#ifdef PBC_HOSTSIM
#define PBC_PRIVATE
#else
#define PBC_PRIVATE __attribute__((address_space(5)))     // the lane's private memory: scratch_load, not flat_load
#endif
template <int N> constexpr bool kLimbFormA = N == 16;     // widths with an instantiated limb-form kernel
template <int N> __shared__ uint32_t g_lds_al[(4 * Limbs29<N>::L + 6) * AL_LANES];

// value bounds are statements about q < 2^(29 L - 10); the host enables the kernel only then (host_params.h)
template <int N>
struct AL {
  static constexpr int L = Limbs29<N>::L;
  static constexpr uint32_t MASK = Limbs29<N>::MASK;
  static constexpr bool kDigits = false;                    // (pairing_aw.cuh: the Solinas loop; AG<N> is the other choice)
  static_assert(Limbs29<N>::W == 29 && 2 * L + L <= 63, "limb-form type a needs 29-bit limbs and room for a doubled operand");
  typedef fl<N> el;
  typedef uint32_t vL __attribute__((ext_vector_type(L)));
  typedef uint32_t vLo __attribute__((ext_vector_type(12)));
  enum { SLOT_FX = 0, SLOT_FY = 1, SLOT_Z = 2, SLOT_ZZ = 3 };
  enum { K2 = 0, K4 = 1, K8 = 2, K12 = 3, K16 = 4 };   // AConst::ksub: (c, D) = (2, 1), (4, 2), (8, 4), (12, 2), (16, 2)

  // ---- host mirror only: worst-case bound tracker ----------------------------------------------------------------
  // Every element carries (u, B) as derived from the operations that produced it -- not from its actual limbs --
  // and every operation asserts its precondition on those, so one run of the mirror proves the bounds for ALL inputs.
#ifdef PBC_HOSTSIM
  static constexpr double U_STRICT = 1.0 - 1.0 / 536870912.0, U_ALMOST = 1.0 + 7.0 / 536870912.0;
  static constexpr double KC[5] = {2, 4, 8, 12, 16}, KD[5] = {1, 2, 4, 2, 2};
  static int leff_() { return L; }                          // (pairing_aw.cuh's mirror: every limb is filled, R / q >= 2^10)
  static double hs_slack() { return 1024.0; }
  static inline double hs_lu[4] = {1, 1, 1, 1}, hs_lB[4] = {1, 1, 1, 1}, hs_au, hs_aB, hs_bu, hs_bB;
  static void hs_fail(const char *what, double v) { fprintf(stderr, "hostsim: limb-form type a: %s (%g)\n", what, v); abort(); }
  static void hs_limbs(const el &a) {                       // the actual limbs must respect the tracked bound
    for (int i = 0; i < L; i++)
      if ((double) a.l[i] > a.hs_u * 536870912.0) hs_fail("limb above its tracked bound", a.hs_u);
  }
  static void hs_set(el &r, double u, double B) { r.hs_u = u; r.hs_B = B; hs_limbs(r); }
  static void hs_dom(const el &b, int k) {                  // K_k dominates b
    hs_limbs(b);
    if (b.hs_u > KD[k] * U_STRICT + 1e-12) hs_fail("subtrahend limbs not dominated", b.hs_u);
    // top limb: K's is at least c q / 2^493 - 1 - D, b's at most B q / 2^493, and q >= 2^504 (init checks it)
    if ((KC[k] - b.hs_B) * 2048.0 < KD[k] + 1) hs_fail("subtrahend value not dominated", b.hs_B);
  }
  static void hs_cols(double s) { if (s > 2.55) hs_fail("column capacity", s); }
#define AL_HS(...) __VA_ARGS__
#else
#define AL_HS(...)
#endif

  static PBC_DEV uint32_t &lds(int slot, int limb) { return g_lds_al<N>[(slot * L + limb) * AL_LANES + threadIdx.x]; }
  static PBC_DEV uint32_t &lds_hi(int i) { return g_lds_al<N>[(4 * L + i) * AL_LANES + threadIdx.x]; }
  static PBC_DEV void lds_get(el &r, int slot) {
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = lds(slot, i);
    AL_HS(hs_set(r, hs_lu[slot], hs_lB[slot]);)
  }
  static PBC_DEV void lds_put(int slot, const el &a) {
#pragma unroll
    for (int i = 0; i < L; i++) lds(slot, i) = a.l[i];
    AL_HS(hs_limbs(a); hs_lu[slot] = a.hs_u; hs_lB[slot] = a.hs_B;)
  }
  static PBC_DEV vL to_v(const el &a) {
    vL v;
#pragma unroll
    for (int i = 0; i < L; i++) v[i] = a.l[i];
    return v;
  }
  static PBC_DEV void from_v(el &r, vL v) {
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = v[i];
  }

  // ---- carry-free additive layer -------------------------------------------------------------------------------
  static PBC_DEV void add(el &r, const el &a, const el &b) {
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = a.l[i] + b.l[i];
    AL_HS(if (a.hs_u + b.hs_u >= 8) hs_fail("sum overflows 32 bits", a.hs_u + b.hs_u); hs_set(r, a.hs_u + b.hs_u, a.hs_B + b.hs_B);)
  }
  template <int S>
  static PBC_DEV void shl(el &r, const el &a) {
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = a.l[i] << S;
    AL_HS(if (a.hs_u * (1 << S) >= 8) hs_fail("shift overflows 32 bits", a.hs_u); hs_set(r, a.hs_u * (1 << S), a.hs_B * (1 << S));)
  }
  // r = a + K - b: b's limbs must be dominated by K's (see the table at ksub in host_params.h)
  static PBC_DEV void subk(el &r, const el &a, const el &b, int k) {
    const uint32_t *K = c_a.ksub[k];
    AL_HS(hs_dom(b, k); const double u = a.hs_u + KD[k] + 1, B = a.hs_B + KC[k]; if (u >= 8) hs_fail("difference overflows 32 bits", u);)
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = a.l[i] - b.l[i] + K[i];
    AL_HS(hs_set(r, u, B);)
  }
  // r = K - b  (= -b mod q)
  static PBC_DEV void negk(el &r, const el &b, int k) {
    const uint32_t *K = c_a.ksub[k];
    AL_HS(hs_dom(b, k);)
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = K[i] - b.l[i];
    AL_HS(hs_set(r, KD[k] + 1, KC[k]);)
  }
  // parallel carry pass: limbs < 2^32 in, limbs <= 2^29 + 6 out (the top limb is not cut: values stay far below 2^522)
  static PBC_DEV void norm(el &r, const el &a) {
    uint32_t c = 0;
    AL_HS(hs_limbs(a); const double B = a.hs_B; if (a.hs_u >= 8) hs_fail("normalising limbs above 32 bits", a.hs_u);)
#pragma unroll
    for (int i = 0; i < L; i++) {
      const uint32_t t = a.l[i];
      r.l[i] = (i < L - 1 ? (t & MASK) : t) + c;
      c = t >> 29;
    }
    AL_HS(hs_set(r, U_ALMOST, B);)
  }
  static PBC_DEV void to_el(el &r, const fp<N> &a) {
    to_limbs<N>(r, a);
    AL_HS(hs_set(r, U_STRICT, 1.0);)
  }
  static PBC_DEV void to_words(fp<N> &r, const el &a) {     // value < 2q (from_limbs subtracts q at most once)
    AL_HS(hs_limbs(a); if (a.hs_u > U_STRICT || a.hs_B >= 2) hs_fail("from_limbs needs a P-class value", a.hs_B);)
    from_limbs<N>(r, a);
  }

  // ---- products (inline bodies) ----------------------------------------------------------------------------------
  // DBL = 1: one of the operands has limbs up to 2^30 (a sum of two, or a doubled value)
  template <int DBL>
  static PBC_DEV void mul_inl(el &r, const el &a, const el &b) {
    const el x[1] = {a}, y[1] = {b};
    AL_HS(hs_limbs(a); hs_limbs(b); hs_cols(a.hs_u * b.hs_u); const double B = 1 + a.hs_B * b.hs_B / 1024;)
    sop_limbs<N, 1, DBL>(r, x, y);
    AL_HS(hs_set(r, U_STRICT, B);)
  }
  static PBC_DEV void sqr_inl(el &r, const el &a) {
    AL_HS(const el x[1] = {a}; hs_limbs(a); hs_cols(a.hs_u * a.hs_u); hs_sop_check<N>(x, x, 1); const double B = 1 + a.hs_B * a.hs_B / 1024;)
    sqr_limbs<N>(r.l, a.l);
    AL_HS(hs_set(r, U_STRICT, B);)
  }
  // sop_limbs<N, 2> without the loop over terms (three loop levels defeat the unroller: the operands end up in scratch)
  static PBC_DEV void sop2_limbs(el &r, const el &a0, const el &b0, const el &a1, const el &b1) {
    const FpK<N> &K = fpk<N>();
    PBC_COUNT_MACS(3 * L * L);
    uint32_t m[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc += (uint64_t) a0.l[i] * b0.l[k - i];
        acc += (uint64_t) a1.l[i] * b1.l[k - i];
      }
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t) m[i] * K.p29[k - i];
      m[k] = ((uint32_t) acc * K.ninv29) & MASK;
      acc += (uint64_t) m[k] * K.p29[0];
      acc >>= 29;
    }
#pragma unroll
    for (int k = L; k < 2 * L; k++) {
#pragma unroll
      for (int i = k - L + 1; i < L; i++) {
        acc += (uint64_t) a0.l[i] * b0.l[k - i];
        acc += (uint64_t) a1.l[i] * b1.l[k - i];
      }
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (uint64_t) m[i] * K.p29[k - i];
      r.l[k - L] = (uint32_t) acc & MASK;
      acc >>= 29;
    }
  }
  // r = a0 b0 + a1 b1, one reduction (all four operands normalised)
  static PBC_DEV void sop2_inl(el &r, const el &a0, const el &b0, const el &a1, const el &b1) {
    AL_HS(const el x[2] = {a0, a1}; const el y[2] = {b0, b1}; hs_limbs(a0); hs_limbs(b0); hs_limbs(a1); hs_limbs(b1); hs_cols(a0.hs_u * b0.hs_u + a1.hs_u * b1.hs_u);
          const double B = 1 + (a0.hs_B * b0.hs_B + a1.hs_B * b1.hs_B) / 1024;)
    AL_HS(hs_sop_check<N>(x, y, 2);)
    sop2_limbs(r, a0, b0, a1, b1);
    AL_HS(hs_set(r, U_STRICT, B);)
  }

  // ---- out-of-line instances -------------------------------------------------------------------------------------
  // (host mirror: the bounds of register operands cross the call in hs_au / hs_aB and friends)
  static __device__ __noinline__ vL mul_fn(vL va, vLo blo) {
    el a, b, r;
    from_v(a, va);
    AL_HS(a.hs_u = hs_au; a.hs_B = hs_aB; b.hs_u = hs_bu; b.hs_B = hs_bB;)
#pragma unroll
    for (int i = 0; i < 12; i++) b.l[i] = blo[i];
#pragma unroll
    for (int i = 12; i < L; i++) b.l[i] = lds_hi(i - 12);
    mul_inl<1>(r, a, b);
    AL_HS(hs_au = r.hs_u; hs_aB = r.hs_B;)
    return to_v(r);
  }
  static __device__ __noinline__ vL sqr_fn(vL va) {
    el a, r;
    from_v(a, va);
    AL_HS(a.hs_u = hs_au; a.hs_B = hs_aB;)
    sqr_inl(r, a);
    AL_HS(hs_au = r.hs_u; hs_aB = r.hs_B;)
    return to_v(r);
  }
  // fx p + (+-fy) q with f read from its LDS slots, one reduction
  static __device__ __noinline__ vL fsop_fn(vL vp, vLo qlo, int neg) {
    el fx, fy, p, q, r;
    from_v(p, vp);
    AL_HS(p.hs_u = hs_au; p.hs_B = hs_aB; q.hs_u = hs_bu; q.hs_B = hs_bB;)
#pragma unroll
    for (int i = 0; i < 12; i++) q.l[i] = qlo[i];
#pragma unroll
    for (int i = 12; i < L; i++) q.l[i] = lds_hi(i - 12);
    lds_get(fx, SLOT_FX);
    lds_get(fy, SLOT_FY);
    if (neg) {
      negk(fy, fy, K2);                // u 2, B 2
      norm(fy, fy);
    }
    sop2_inl(r, fx, p, fy, q);
    AL_HS(hs_au = r.hs_u; hs_aB = r.hs_B;)
    return to_v(r);
  }
  static PBC_DEV void fsop(el &r, const el &p, const el &q, int neg) {
    vLo lo;
#pragma unroll
    for (int i = 0; i < 12; i++) lo[i] = q.l[i];
#pragma unroll
    for (int i = 12; i < L; i++) lds_hi(i - 12) = q.l[i];
    AL_HS(hs_au = p.hs_u; hs_aB = p.hs_B; hs_bu = q.hs_u; hs_bB = q.hs_B;)
    from_v(r, fsop_fn(to_v(p), lo, neg));
    AL_HS(r.hs_u = hs_au; r.hs_B = hs_aB;)
  }

  static PBC_DEV void mul(el &r, const el &a, const el &b) {
    vLo lo;
#pragma unroll
    for (int i = 0; i < 12; i++) lo[i] = b.l[i];
#pragma unroll
    for (int i = 12; i < L; i++) lds_hi(i - 12) = b.l[i];
    AL_HS(hs_au = a.hs_u; hs_aB = a.hs_B; hs_bu = b.hs_u; hs_bB = b.hs_B;)
    from_v(r, mul_fn(to_v(a), lo));
    AL_HS(r.hs_u = hs_au; r.hs_B = hs_aB;)
  }
  static PBC_DEV void muls(el &r, const el &a, int slot) {           // second operand from an LDS slot
    el b;
    lds_get(b, slot);
    mul(r, a, b);
  }
  static PBC_DEV void sqr(el &r, const el &a) {
    AL_HS(hs_au = a.hs_u; hs_aB = a.hs_B;)
    from_v(r, sqr_fn(to_v(a)));
    AL_HS(r.hs_u = hs_au; r.hs_B = hs_aB;)
  }
  // f <- f^2 in LDS  (fi_square, fieldquadratic.c:459-477: (x + y)(x - y) + 2xy i)
  static PBC_DEV void fsqr() {
    el fx, fy, e0, e1, r;
    lds_get(fx, SLOT_FX);
    lds_get(fy, SLOT_FY);
    add(e0, fx, fy);                   // u 2
    subk(e1, fx, fy, K2);              // u 3, B 3.5
    norm(e1, e1);
    mul(r, e0, e1);
    lds_put(SLOT_FX, r);
    shl<1>(e0, fx);                    // u 2
    mul(r, e0, fy);
    lds_put(SLOT_FY, r);
  }
  // f <- f (lx + i ly) in LDS  (fi_mul, fieldquadratic.c:425-457) as two lazy sums of two products:
  //     re = fx lx + (-fy) ly,   im = fx ly + fy lx        -- the multiply-adds of Karatsuba's three products, no
  // additions besides the negation.  lx, ly: normalised, B <= 6.
  static PBC_DEV void fmul(const el &lx, const el &ly) {
    el re, im;
    fsop(re, lx, ly, 1);
    fsop(im, ly, lx, 0);
    lds_put(SLOT_FX, re);
    lds_put(SLOT_FY, im);
  }

  // Where the second pairing argument Q = (Qx, Qy) waits: the lane's private memory, canonical limbs.  (A product kernel
  // on these routines with the per-term state in global memory, as a_prod_pairing_lane keeps it, was measured at
  // 0.875 M against 0.895 M products/s for the word-form one -- 25 % more workspace traffic -- and is not in the tree.)
  // (volatile: the compiler would otherwise hoist these loop-invariant loads out of the Miller loop into 36 registers
  // and spill those -- 170 scratch reloads per step instead of 36 reads)
  // (PBC_A_QVEC 1: each coordinate padded to QW words and read with 16-byte loads -- 10 scratch instructions per step
  // instead of 36)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  static constexpr int QW = PBC_A_QVEC ? (L + 3) / 4 * 4 : L;
  struct QPriv {
    const volatile PBC_PRIVATE uint32_t *p;
    PBC_DEV void get(el &r, int which) const {
      if constexpr (PBC_A_QVEC != 0) {
        const volatile PBC_PRIVATE u32x4 *p4 = (const volatile PBC_PRIVATE u32x4 *) (p + which * QW);
#pragma unroll
        for (int i = 0; i < QW / 4; i++) {
          const u32x4 t = p4[i];
#pragma unroll
          for (int c = 0; c < 4; c++) if (4 * i + c < L) r.l[4 * i + c] = t[c];
        }
      } else {
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = p[which * L + i];
      }
      AL_HS(hs_set(r, U_STRICT, 1.0);)
    }
  };

  // The point V = (X, Y, Z) of the Miller loop: X, Y in registers (almost normalised, B <= 14), Z and Z^2 in their
  // LDS slots (P-class); Q = (Qx, Qy) in the lane's private memory as 2 x 18 canonical limbs.
  struct jacl {
    el X, Y;
  };

  // One doubling step: f <- f^2 l_{V,V}(phi(Q)), V <- 2V.  Same line as a_double_step (pairing_a.cuh):
  //     re = M (ZZ Qx + X) - 2 Y^2,  im = (2YZ) ZZ Qy,  M = 3X^2 + Z^4.
  // Where the word-form step trades products for squarings, this one takes products: a squaring of a sum needs the
  // sum normalised first, which costs more than the squaring saves.  9 M + 6 S + 2 two-term sums per step.
  static PBC_DEV void double_step(jacl &V, const QPriv &Q) {
    el XX, YY, M, t0, t1, lx, ly, Z3, S1, W;
    pbc_fair_tick<PBC_A_FAIR_BIT>();
    fsqr();
    sqr(XX, V.X);
    lds_get(t0, SLOT_ZZ);
    sqr(t0, t0);                       // Z^4
    shl<1>(M, XX);
    add(M, M, XX);
    add(M, M, t0);                     // u 4, B 5
    norm(M, M);
    sqr(YY, V.Y);
    Q.get(t1, 0);
    muls(t0, t1, SLOT_ZZ);             // ZZ Qx
    add(t0, t0, V.X);                  // u 2, B 10.5
    mul(lx, M, t0);                    // P
    shl<1>(t1, YY);                    // u 2, B 3
    subk(lx, lx, t1, K4);              // u 4, B 5.5
    norm(lx, lx);
    shl<1>(t1, V.Y);                   // u 2
    muls(Z3, t1, SLOT_Z);              // Z3 = 2YZ
    muls(t1, Z3, SLOT_ZZ);             //                      (Z, ZZ dead)
    lds_put(SLOT_Z, Z3);
    sqr(Z3, Z3);
    lds_put(SLOT_ZZ, Z3);
    Q.get(Z3, 1);
    mul(ly, t1, Z3);                   // im = Z3 ZZ Qy
    fmul(lx, ly);
    shl<1>(t1, V.X);                   // u 2
    mul(S1, YY, t1);                   // 2XY^2
    sqr(t0, M);
    shl<2>(t1, S1);                    // 8XY^2: u 4, B 6
    subk(t0, t0, t1, K8);              // u 6, B 9.5
    norm(V.X, t0);                     // X3 = M^2 - 2S
    shl<1>(t1, S1);                    // S = 4XY^2: u 2, B 3
    subk(W, t1, V.X, K12);             // S - X3: u 5, B 15
    norm(W, W);
    mul(t0, M, W);
    sqr(t1, YY);                       // Y^4
    shl<3>(t1, t1);                    // u 8, B 12
    norm(t1, t1);
    subk(t0, t0, t1, K12);             // u 4, B 13.1
    norm(V.Y, t0);                     // Y3 = M (S - X3) - 8Y^4
  }

  // Mixed addition step (a_add_step, pairing_a.cuh): runs once per pairing, so every difference is normalised at once.
  static PBC_DEV void add_step(jacl &V, const el &x2, const el &y2, const QPriv &Q) {
    el H, R, Z3, HH, HHH, t0, t1, lx, ly;
    muls(t0, x2, SLOT_ZZ);
    subk(H, t0, V.X, K16);
    norm(H, H);                        // B 17.5
    lds_get(t0, SLOT_Z);
    muls(t0, t0, SLOT_ZZ);
    mul(t0, y2, t0);
    subk(R, t0, V.Y, K16);
    norm(R, R);                        // B 17.5
    muls(Z3, H, SLOT_Z);
    Q.get(t0, 0);
    add(t0, t0, x2);                   // u 2
    mul(lx, R, t0);
    mul(t0, Z3, y2);
    subk(lx, lx, t0, K2);
    norm(lx, lx);                      // B 3.5
    Q.get(t0, 1);
    mul(ly, Z3, t0);
    sqr(HH, H);
    mul(HHH, HH, H);
    mul(t0, V.X, HH);                  // X1 H^2
    sqr(t1, R);
    subk(t1, t1, HHH, K2);
    norm(t1, t1);                      // B 3.5
    shl<1>(HH, t0);                    // u 2, B 3
    subk(t1, t1, HH, K4);
    norm(t1, t1);                      // X3 = R^2 - H^3 - 2 X1 H^2, B 7.5
    subk(t0, t0, t1, K16);
    norm(t0, t0);                      // B 17.5
    mul(t0, R, t0);
    mul(HHH, V.Y, HHH);
    subk(t0, t0, HHH, K2);
    norm(V.Y, t0);                     // Y3 = R (X1 H^2 - X3) - Y1 H^3, B 3.5
    V.X = t1;
    lds_put(SLOT_Z, Z3);
    sqr(Z3, Z3);
    lds_put(SLOT_ZZ, Z3);
    fmul(lx, ly);
  }

  // limb form of any class -> fully reduced words: one product by R mod q brings the value below 2q
  static PBC_DEV void to_fp(fp<N> &r, const el &a) {
    el one, t;
    fp<N> w;
    fp_set<N>(w, fpk<N>().one);
    to_el(one, w);
    mul(t, a, one);
    to_words(r, t);
  }

  // f^((q^2-1)/r) with f in LDS: a_final_exp (pairing_a.cuh) with the Lucas ladder in limb form.  Output: words.
  static PBC_DEV void final_exp(fp2<N> &out) {
    el fx, fy, a2, b2, Nn, A, B, t, w, g0, P, v0, v1, two;
    fp<N> tw;
    lds_get(fx, SLOT_FX);
    lds_get(fy, SLOT_FY);
    sqr(a2, fx);
    sqr(b2, fy);
    add(Nn, a2, b2);
    norm(Nn, Nn);                      // B 3
    subk(A, a2, b2, K2);
    norm(A, A);                        // B 3.5
    shl<1>(t, fx);
    mul(B, t, fy);                     // 2 fx fy
    negk(B, B, K2);
    norm(B, B);                        // f^(q-1) = (A + B i) / N
    {                                  // B = 0 (f^(q-1) = +-1): invert N * 1 instead, see a_final_exp
      fp<N> bw, one;
      to_fp(bw, B);
      fp_set<N>(one, fpk<N>().one);
      const bool z = fp_is0<N>(bw);
      fp_cmov<N>(bw, one, z);
      to_el(B, bw);                    // canonical from here on
    }
    mul(t, Nn, B);
    to_words(tw, t);
    fp_inv<N>(tw, tw);                 // 1/(N B)
    to_el(t, tw);
    mul(w, t, B);                      // 1/N
    mul(g0, A, w);                     // Re f^(q-1)
    mul(t, t, Nn);                     // 1/B
    mul(w, t, Nn);                     // N/B = 1/Im f^(q-1)
    shl<1>(P, g0);
    norm(P, P);                        // B 3, almost normalised
    fp_set<N>(tw, fpk<N>().one);
    fp_dbl<N>(tw, tw);
    to_el(two, tw);                    // canonical
    v0 = two;
    v1 = P;
    for (int j = c_a.hbits - 1; j >= 0; j--) {
      const bool bit = j ? ((c_a.h[j >> 5] >> (j & 31)) & 1) : false;
      el m, s;
      if ((j & 15) == 0) pbc_fair_tick<PBC_A_FAIR_BIT>();
      mul(m, v0, v1);
      subk(m, m, P, K4);               // P almost normalised, B 2.1
      norm(m, m);                      // B 5.1
      if (bit) {
        sqr(s, v1);
        subk(s, s, two, K2);
        norm(v1, s);                   // B 3.5
        v0 = m;
      } else {
        sqr(s, v0);
        subk(s, s, two, K2);
        norm(v0, s);
        v1 = m;
      }
    }
    // out.y = -(2 v1 - P v0) (N/B) / 4,  out.x = v0 / 2
    mul(t, v0, P);
    shl<1>(v1, v1);                    // u 2, B 35
    subk(v1, v1, t, K2);
    norm(v1, v1);                      // B 37
    mul(v1, v1, w);
    fp<N> y, x;
    to_words(y, v1);
    fp_halve<N>(y, y);
    fp_halve<N>(y, y);
    fp_neg<N>(out.y, y);
    to_fp(x, v0);
    fp_halve<N>(out.x, x);
  }

  // wave-uniform words (the table of a preprocessed first argument) -> limbs, written with plain shifts so that the
  // conversion stays on the scalar unit
  static PBC_DEV void to_el_uniform(el &r, const uint32_t *w) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      const int bit = 29 * i, j = bit >> 5, sh = bit & 31;
      const uint64_t pair = ((uint64_t) (j + 1 < N ? w[j + 1] : 0u) << 32) | w[j];
      r.l[i] = (uint32_t) (pair >> sh) & MASK;
    }
    AL_HS(hs_set(r, U_STRICT, 1.0);)
  }
  // f <- f * ((cA Qx + cC) + i cB Qy) for table entry idx (a_pairing_pp_apply's step, ecc/a_param.c:317-360)
  static PBC_DEV void pp_line(const uint32_t *tab, int idx, const el &Qx, const el &Qy) {
    el c, lx, ly;
    to_el_uniform(c, tab + (idx * 3 + 0) * N);
    mul(lx, Qx, c);
    to_el_uniform(c, tab + (idx * 3 + 2) * N);
    add(lx, lx, c);                    // u 2, B 2.5
    norm(lx, lx);
    to_el_uniform(c, tab + (idx * 3 + 1) * N);
    mul(ly, Qy, c);
    fmul(lx, ly);
  }
  // pairing_pp_apply for one lane (the table is the word-form one a_pp_init_lane writes, pairing_a.cuh).  Q stays in
  // registers here: nothing else competes for them.
  static PBC_DEV void pp_apply_lane(uint8_t *gt, const uint32_t *tab, bool p_valid, const uint8_t *g2) {
    constexpr int NB = 4 * N;
    el Qx, Qy;
    bool valid;
    {
      fp<N> x, y;
      el q;
      fp_load_be<N>(x, g2);
      fp_load_be<N>(y, g2 + NB);
      valid = p_valid & a_on_curve<N>(x, y);
      to_el(Qx, x);
      to_el(Qy, y);
      fp_set<N>(x, fpk<N>().one);
      to_el(q, x);
      lds_put(SLOT_FX, q);
#pragma unroll
      for (int i = 0; i < L; i++) q.l[i] = 0;
      lds_put(SLOT_FY, q);
    }
    int slot = 0;
    for (int i = c_a.exp2 - 1; i >= 0; i--, slot++) {
      pbc_fair_tick<PBC_A_FAIR_BIT>();
      fsqr();
      pp_line(tab, slot, Qx, Qy);
      if (i == c_a.exp1) pp_line(tab, c_a.exp2, Qx, Qy);
    }
    fp2<N> out;
    final_exp(out);
    a_store_gt<N>(gt, out, valid);
  }

  // The Miller loop of one (P, Q) pair: f ends up in its LDS slots (P-class), the return value says whether both
  // arguments were acceptable (on the curve, P not of order two / O)
  static PBC_DEV bool miller_lane(const uint8_t *g1, const uint8_t *g2) {
    constexpr int NB = 4 * N;
    jacl V;
    __attribute__((aligned(16))) uint32_t Qm[2 * QW];      // Q in private memory: read twice per step, by address
    bool valid;
    {
      fp<N> Px, Py, Qx, Qy;
      el q;
      fp_load_be<N>(Px, g1);
      fp_load_be<N>(Py, g1 + NB);
      fp_load_be<N>(Qx, g2);
      fp_load_be<N>(Qy, g2 + NB);
      valid = a_first_arg_ok<N>(Px, Py) & a_on_curve<N>(Qx, Qy);
      to_el(q, Qx);
#pragma unroll
      for (int i = 0; i < L; i++) Qm[i] = q.l[i];
      to_el(q, Qy);
#pragma unroll
      for (int i = 0; i < L; i++) Qm[QW + i] = q.l[i];
#pragma unroll
      for (int i = L; i < QW; i++) Qm[i] = Qm[QW + i] = 0;
      to_el(V.X, Px);
      to_el(V.Y, Py);
      fp_set<N>(Px, fpk<N>().one);
      to_el(q, Px);
      lds_put(SLOT_Z, q);
      lds_put(SLOT_ZZ, q);
      lds_put(SLOT_FX, q);
#pragma unroll
      for (int i = 0; i < L; i++) q.l[i] = 0;
      lds_put(SLOT_FY, q);
    }
    const QPriv Q = {(const volatile PBC_PRIVATE uint32_t *) Qm};
    for (int i = c_a.exp2 - 1; i >= 0; i--) {
      double_step(V, Q);
      if (i == c_a.exp1) {             // the one non-zero middle digit of r: V <- V +- P
        fp<N> Px, Py;                  // (P is read again here rather than kept in 36 registers across the loop)
        el x2, y2;
        fp_load_be<N>(Px, g1);
        fp_load_be<N>(Py, g1 + NB);
        to_el(x2, Px);
        to_el(y2, Py);
        if (c_a.sign1 < 0) {
          negk(y2, y2, K2);
          norm(y2, y2);
        }
        add_step(V, x2, y2, Q);
      }
    }
    return valid;
  }

  // element_pairing for one lane
  static PBC_DEV void pairing_lane(uint8_t *gt, const uint8_t *g1, const uint8_t *g2) {
    const bool valid = miller_lane(g1, g2);
    fp2<N> out;
    final_exp(out);
    a_store_gt<N>(gt, out, valid);
  }

  // ---- element_prod_pairing with one TERM per lane -----------------------------------------------------------------
  // The reference's a_pairings_affine (ecc/a_param.c:1283-1383) shares the accumulator's squaring between the k terms
  // of a product; one lane running k terms that way (a_prod_pairing_lane, pairing_a.cuh) has to keep the per-term
  // state in a global workspace (345 GB of traffic per 2^18 x 16 launch) and runs the word-form routines.  Here every
  // term is a lane of the single-pairing Miller loop above (limb form, nothing outside registers and LDS; 12 % more
  // multiply-adds per term for its own squaring of f), which leaves its Miller value in a 160-byte workspace record;
  // a second, short kernel multiplies the k values of a product and runs ONE final exponentiation (the F_q^* factors
  // by which the Miller values differ from the reference's die there).
  static constexpr int MREC = (2 * L + 1 + 3) / 4;        // uint4s per record: fx, fy (L limbs each), flag
  static PBC_DEV void miller_record_lane(uint4 *rec, const uint8_t *g1, const uint8_t *g2) {
    const bool valid = miller_lane(g1, g2);
    el fx, fy;
    lds_get(fx, SLOT_FX);
    lds_get(fy, SLOT_FY);
    AL_HS(if (fx.hs_u > U_STRICT || fy.hs_u > U_STRICT || fx.hs_B > 1.5 || fy.hs_B > 1.5) hs_fail("Miller value not P-class", fx.hs_B);)
    uint32_t w[4 * MREC];
#pragma unroll
    for (int i = 0; i < L; i++) { w[i] = fx.l[i]; w[L + i] = fy.l[i]; }
#pragma unroll
    for (int i = 2 * L; i < 4 * MREC; i++) w[i] = valid ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < MREC; i++) rec[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
  }
  static PBC_DEV bool record_get(el &x, el &y, const uint4 *rec) {
    uint32_t w[4 * MREC];
#pragma unroll
    for (int i = 0; i < MREC; i++) {
      const uint4 t = rec[i];
      w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < L; i++) { x.l[i] = w[i]; y.l[i] = w[L + i]; }
    AL_HS(hs_set(x, U_STRICT, 1.5); hs_set(y, U_STRICT, 1.5);)
    return w[2 * L] != 0;
  }
  // one product per lane: the k records of its terms -> GT bytes (any unacceptable term: the identity, as the other
  // product kernels and element_prod_pairing's callers see it)
  static PBC_DEV void prod_finish_lane(uint8_t *gt, const uint4 *rec, int k) {
    el x, y;
    bool valid = record_get(x, y, rec);
    lds_put(SLOT_FX, x);
    lds_put(SLOT_FY, y);
    for (int j = 1; j < k; j++) {
      valid &= record_get(x, y, rec + (size_t) j * MREC);
      fmul(x, y);
    }
    fp2<N> out;
    final_exp(out);
    a_store_gt<N>(gt, out, valid);
  }
};

}  // namespace pbc
