// fp.cuh -- fixed-width, register-resident Montgomery arithmetic in F_q for gfx950.
//
// Replaces the reference's GMP-backed arith/montfp.c (mont_mul :334-364, fp_add/sub/
// double/halve/neg :220-330, fp_invert :401-422, fp_to/from_bytes :487-517) with
// N x 32-bit limbs held in VGPRs, one field element per lane.  The modulus and the
// Montgomery constants are wave-uniform (kernel argument segment -> scalar loads -> SGPRs).
//
// Representation: N little-endian 32-bit words per element, Montgomery form, always fully
// reduced to [0, q).  The Montgomery radix is R = 2^(29 L), L = ceil(32N/29), because the
// multiplier works on 29-bit limbs (see fp_mul29_inl); the radix is private: all exchange
// with the host / the reference is in canonical big-endian bytes (SURVEY.md "Key facts").
#pragma once
#ifdef PBC_HOSTSIM
#include "hostsim_shim.h"   // tests/hostsim: the same source compiled for the CPU (debug mirror)
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

// Executed-work accounting: every multiplier body reports its multiply-adds.  A no-op in the library; the host
// mirror (tests/hostsim) defines the hook to count, which is where bench.py's executed_macs_per_unit comes from.
#ifndef PBC_COUNT_MACS
#define PBC_COUNT_MACS(n) ((void) 0)
#endif

namespace pbc {

template <int N>
struct FpK {            // wave-uniform field constants (field_init_mont_fp, montfp.c:533-600)
  uint32_t p[N];        // modulus q
  uint32_t one[N];      // R mod q
  uint32_t r2[N];       // R^2 mod q
  uint32_t pm2[N];      // q - 2 (Fermat exponent)
  uint32_t ninv;        // -q^-1 mod 2^32   (32-bit-limb product scanning, variant 0)
  uint32_t pbits;       // bit length of q
  uint32_t p29[(32 * N + 27) / 28];   // q in 29-bit limbs (28-bit for N >= 32; unsaturated multiplier)
  uint32_t ninv29;      // -q^-1 mod 2^29 (2^28)
  uint32_t r3[N];       // R^3 mod q (puts an inverse back into Montgomery form, cf. montfp.c:417)
  uint32_t p30[(32 * N + 2 + 29) / 30];   // q in 30-bit limbs (safegcd inversion)
  uint32_t qinv30;      // q^-1 mod 2^30
  uint32_t fbytes;      // fixed_length_in_bytes = ceil(bits(q)/8) (montfp.c:577); <= 4N
};

template <int N>
struct fp {
  uint32_t v[N];
};

#define PBC_DEV __device__ __forceinline__

// Per-object constants travel in the KERNEL ARGUMENT SEGMENT.  Every kernel of the library takes the constant block
// of its pairing object (KArgs<N>, below) as its LAST argument, by value.  The hidden arguments of a HIP kernel follow
// the explicit ones, and the ABI hands their address (the "implicit argument pointer") down to callees of any depth in
// SGPRs -- so device code finds the block at a fixed negative offset from that pointer.  The segment is constant
// address space: every read is a scalar load from a wave-uniform address (-> SGPR operands of the multiply-adds),
// exactly as a __constant__ symbol would give, but nothing is process-global: two pairing objects (or one object on two
// streams) cannot disturb each other, and a launch needs no upload.
// Layout of the block, from its END backwards (the last three parts do not depend on the field width):
//     end - 2560 + [0, 704)      CurveK    curve coefficients, cofactor, square-root recipe      (group_ops.cuh)
//     end - 2560 + [704, 2192)   AConst / DConst / FConst / EConst   the pairing family's constants  (pairing_*.cuh)
//     end - 2560 + [2192, 2560)  ExtSqrtK  square roots in the field of the G2 twist             (group_ops.cuh)
//     end - sizeof(KArgs<N>)     FpK<N>    modulus and Montgomery constants
// Word counts built into the library: 5/6/7 words = the 149..224-bit MNT, Freeman and BN fields of
// the shipped type d / g / f parameter files, 8 words = 256-bit BN fields (type f), 16 words = the
// 512-bit type a field, 33 words = the 1033-bit type a1 field.
#define PBC_FOR_EACH_N(X) X(5) X(6) X(7) X(8) X(16) X(33)
constexpr int KOFF_CURVE = 0, KOFF_TYPE = 704, KOFF_XS = 2192, KOFF_END = 2560;
constexpr int KOFF_OPT = KOFF_END - 4;    // the last word of the block: per-launch options (bit 0: no time-sliced priorities, pbc_fair_tick)
template <int N>
struct alignas(16) KArgs {
  FpK<N> fp;
  alignas(16) uint8_t head[KOFF_END];
};
static_assert(sizeof(KArgs<33>) + 384 <= 4096, "the constant block must fit the kernel argument segment");
static_assert(sizeof(KArgs<5>) % 16 == 0 && sizeof(KArgs<16>) % 16 == 0, "the block must end where the hidden arguments begin");
// Scalar loads take unsigned immediate offsets, so the block is addressed from a base 4 KB below the hidden arguments
// (one s_add / s_addc per function; the empty asm keeps the compiler from folding the subtraction back into every
// access, which would cost an address computation per load).
constexpr int KSEG = 4096;
#ifndef PBC_HOSTSIM
PBC_DEV const uint8_t *pbc_kargs_base() {
  uint64_t b = (uint64_t) (const uint8_t *) (const __attribute__((address_space(4))) uint8_t *) __builtin_amdgcn_implicitarg_ptr() - KSEG;
  asm("" : "+s"(b));
  return (const uint8_t *) (const __attribute__((address_space(4))) uint8_t *) b;
}
#endif
template <class T, int OFF>
PBC_DEV const T &kconst() { return *reinterpret_cast<const T *>(pbc_kargs_base() + (KSEG - KOFF_END + OFF)); }
template <int N> PBC_DEV const FpK<N> &fpk() { return *reinterpret_cast<const FpK<N> *>(pbc_kargs_base() + (KSEG - (int) sizeof(KArgs<N>))); }

// Time-sliced fairness between the two waves of a SIMD.  With equal priorities the arbiter prefers the older wave: of two
// waves that start together on identical work (type f, resident workgroups) one ran its two pairings in 21 ms, the other
// in 31 ms -- the last third of the launch with one wave per SIMD, i.e. at half the multiply-add rate (per-wave timestamps,
// profiles/r03_notes.md).  The waves of a SIMD sit in slots of different parity and read the same clock: a wave takes the
// high priority when bit BIT of the clock equals its slot's parity, the low one otherwise.  Called at the entry of
// long-running operations.
template <int BIT>
PBC_DEV void pbc_fair_tick() {
#if !defined(PBC_HOSTSIM) && !defined(PBC_NO_FAIR)
  if (*reinterpret_cast<const uint32_t *>(pbc_kargs_base() + (KSEG - KOFF_END + KOFF_OPT)) & 1u) return;    // "hip_no_fair 1"
  uint32_t hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const uint64_t t = __builtin_readcyclecounter();
  if (((uint32_t) (t >> BIT) ^ hw) & 1) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(0);
#endif
}

// r = (carry:t) >= p ? t - p : t      (final correction of add / mul)
// __builtin_addc/__builtin_subc lower to v_addc_co_u32 / v_subb_co_u32 chains.
template <int N>
PBC_DEV void fp_cond_sub(fp<N> &r, const uint32_t *t, uint32_t carry) {
  const FpK<N> &K = fpk<N>();
  uint32_t d[N];
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < N; i++) d[i] = __builtin_subc(t[i], K.p[i], bw, &bw);
  bool ge = (carry != 0) | (bw == 0);
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = ge ? d[i] : t[i];
}

// ---------------------------------------------------------------------------------------
// Unsaturated multiplier (same value as mont_mul, arith/montfp.c:334-364).  Storage stays N saturated 32-bit
// words (cheap carry-chain add/sub, 32 VGPR arguments per call), but the product is formed on L = ceil(32N/29)
// limbs of 29 bits with R = 2^(29 L):  a column holds at most 2L products < 2^58, so a
// plain 64-bit accumulator never overflows (2L * 2^58 < 2^64 for L <= 31) and every MAC is
// ONE v_mad_u64_u32 -- no carry instruction, no inline asm, free compiler scheduling.
// Measured on MI355X: v_mad_u64_u32 issues every ~4.2 cycles (3.5 with an SGPR factor)
// against 7.5 for the mad + addc pair a saturated 32-bit limb needs (profiles/r01_probe_v1.txt; that first
// version of the multiplier is gone from the source).
// ---------------------------------------------------------------------------------------
// Limb width: 29 bits, except 28 for the 33-word fields -- with L = 37 limbs of 29 bits a column would
// hold up to 74 products of 2^58, which overflows 64 bits for moduli with dense limbs; 38 limbs of 28
// bits (76 x 2^56 = 2^62.3) are safe for every modulus at +5 % multiply-adds.
template <int N>
struct Limbs29 {
  static constexpr int W = N >= 32 ? 28 : 29;
  static constexpr int L = (32 * N + W - 1) / W;
  static constexpr uint32_t MASK = (1u << W) - 1;
  static_assert(2 * L * (1ull << (2 * W - 56)) < 256, "column accumulator would overflow");   // 2L 2^(2W) < 2^64
};

template <int N>
PBC_DEV void to29(uint32_t *l, const fp<N> &a) {
  constexpr int L = Limbs29<N>::L, W = Limbs29<N>::W;
#pragma unroll
  for (int i = 0; i < L; i++) {
    const int bit = W * i, j = bit >> 5, sh = bit & 31;
    uint32_t lo = a.v[j];
    uint32_t x;
    if (sh == 0) x = lo;
    else if (sh + W <= 32 || j + 1 >= N) x = lo >> sh;
    else x = __builtin_amdgcn_alignbit(a.v[j + 1 < N ? j + 1 : j], lo, sh);
    l[i] = (32 * N - bit >= W && sh + W != 32) ? (x & Limbs29<N>::MASK) : x;
  }
}
// L normalised limbs (+ the value may reach 2^(32N)) -> N words + carry word
template <int N>
PBC_DEV uint32_t from29(uint32_t *w, const uint32_t *l) {
  constexpr int L = Limbs29<N>::L, W = Limbs29<N>::W;
#pragma unroll
  for (int j = 0; j <= N; j++) {
    const int bit = 32 * j, i = bit / W, o = bit - W * i;
    uint32_t x = 0;
    if (i < L) x = l[i] >> o;
    if (i + 1 < L) x |= l[i + 1] << (W - o);
    if (2 * W - o < 32 && i + 2 < L) x |= l[i + 2] << (2 * W - o);
    if (j < N) w[j] = x; else return x;
  }
  return 0;
}

// ACC2: split each column over two accumulators (shorter dependent mad chains)
#define PBC_MAC(i_, X_, Y_)                                         \
  do {                                                              \
    if (ACC2 && ((i_) & 1)) acc1 += (uint64_t) (X_) * (Y_);         \
    else acc += (uint64_t) (X_) * (Y_);                             \
  } while (0)

template <int N, bool ACC2>
PBC_DEV void fp_mul29_inl(fp<N> &r, const fp<N> &a, const fp<N> &b) {
  const FpK<N> &K = fpk<N>();
  constexpr int L = Limbs29<N>::L;
  constexpr uint32_t MASK = Limbs29<N>::MASK;
  uint32_t x[L], y[L], m[L], t[L];
  PBC_COUNT_MACS(2 * L * L);
  to29<N>(x, a);
  to29<N>(y, b);
  uint64_t acc = 0, acc1 = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) PBC_MAC(i, x[i], y[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) PBC_MAC(i + 1, m[i], K.p29[k - i]);
    if (ACC2) { acc += acc1; acc1 = 0; }
    m[k] = ((uint32_t) acc * K.ninv29) & MASK;
    acc += (uint64_t) m[k] * K.p29[0];
    acc >>= Limbs29<N>::W;
  }
#pragma unroll
  for (int k = L; k < 2 * L; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) PBC_MAC(i, x[i], y[k - i]);
#pragma unroll
    for (int i = k - L + 1; i < L; i++) PBC_MAC(i + 1, m[i], K.p29[k - i]);
    if (ACC2) { acc += acc1; acc1 = 0; }
    t[k - L] = (uint32_t) acc & MASK;
    acc >>= Limbs29<N>::W;
  }
  uint32_t w[N];
  uint32_t carry = from29<N>(w, t);
  fp_cond_sub<N>(r, w, carry);
}

// Squaring on the same scheme: the cross products are formed once against a pre-doubled copy
// (L(L+1)/2 + L^2 MACs instead of 2 L^2).  Column bound: (L/2) 2^59 + 2^58 + L 2^58 < 2^64.
template <int N, bool ACC2>
PBC_DEV void fp_sqr29_inl(fp<N> &r, const fp<N> &a) {
  const FpK<N> &K = fpk<N>();
  constexpr int L = Limbs29<N>::L;
  constexpr uint32_t MASK = Limbs29<N>::MASK;
  uint32_t x[L], x2[L], m[L], t[L];
  PBC_COUNT_MACS(L * (L + 1) / 2 + L * L);
  to29<N>(x, a);
#pragma unroll
  for (int i = 0; i < L; i++) x2[i] = x[i] << 1;
  uint64_t acc = 0, acc1 = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) PBC_MAC(i, x[i], x2[k - i]);
    if ((k & 1) == 0) PBC_MAC(1, x[k / 2], x[k / 2]);
#pragma unroll
    for (int i = 0; i < k; i++) PBC_MAC(i, m[i], K.p29[k - i]);
    if (ACC2) { acc += acc1; acc1 = 0; }
    m[k] = ((uint32_t) acc * K.ninv29) & MASK;
    acc += (uint64_t) m[k] * K.p29[0];
    acc >>= Limbs29<N>::W;
  }
#pragma unroll
  for (int k = L; k < 2 * L; k++) {
#pragma unroll
    for (int i = k - L + 1; 2 * i < k; i++) PBC_MAC(i, x[i], x2[k - i]);
    if ((k & 1) == 0) PBC_MAC(1, x[k / 2], x[k / 2]);
#pragma unroll
    for (int i = k - L + 1; i < L; i++) PBC_MAC(i, m[i], K.p29[k - i]);
    if (ACC2) { acc += acc1; acc1 = 0; }
    t[k - L] = (uint32_t) acc & MASK;
    acc >>= Limbs29<N>::W;
  }
  uint32_t w[N];
  uint32_t carry = from29<N>(w, t);
  fp_cond_sub<N>(r, w, carry);
}
// r = (a b + c d) / R mod q with ONE reduction: 3 L^2 multiply-adds instead of the 4 L^2 of two products.
// Column bound: 3 L 2^(2W) < 2^64 for both limb widths (114 x 2^56, 54 x 2^58); the value before the final conditional
// subtraction is below q (1 + 2q/R) < 2q.
template <int N, bool ACC2>
PBC_DEV void fp_sop2_29_inl(fp<N> &r, const fp<N> &a, const fp<N> &b, const fp<N> &c, const fp<N> &d) {
  const FpK<N> &K = fpk<N>();
  constexpr int L = Limbs29<N>::L;
  constexpr uint32_t MASK = Limbs29<N>::MASK;
  static_assert(3ull * L * (1ull << (2 * Limbs29<N>::W - 56)) < 256, "column accumulator would overflow");
  uint32_t x[L], y[L], u[L], v[L], m[L], t[L];
  PBC_COUNT_MACS(3 * L * L);
  to29<N>(x, a);
  to29<N>(y, b);
  to29<N>(u, c);
  to29<N>(v, d);
  uint64_t acc = 0, acc1 = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) { PBC_MAC(i, x[i], y[k - i]); PBC_MAC(i + 1, u[i], v[k - i]); }
#pragma unroll
    for (int i = 0; i < k; i++) PBC_MAC(i + 1, m[i], K.p29[k - i]);
    if (ACC2) { acc += acc1; acc1 = 0; }
    m[k] = ((uint32_t) acc * K.ninv29) & MASK;
    acc += (uint64_t) m[k] * K.p29[0];
    acc >>= Limbs29<N>::W;
  }
#pragma unroll
  for (int k = L; k < 2 * L; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) { PBC_MAC(i, x[i], y[k - i]); PBC_MAC(i + 1, u[i], v[k - i]); }
#pragma unroll
    for (int i = k - L + 1; i < L; i++) PBC_MAC(i + 1, m[i], K.p29[k - i]);
    if (ACC2) { acc += acc1; acc1 = 0; }
    t[k - L] = (uint32_t) acc & MASK;
    acc >>= Limbs29<N>::W;
  }
  uint32_t w[N];
  uint32_t carry = from29<N>(w, t);
  fp_cond_sub<N>(r, w, carry);
}
#undef PBC_MAC

// Limb-domain square: limbs in, limbs out (normalised input; same column bound as fp_sqr29_inl)
template <int N>
PBC_DEV void sqr_limbs(uint32_t *t, const uint32_t *x) {
  const FpK<N> &K = fpk<N>();
  constexpr int L = Limbs29<N>::L;
  constexpr uint32_t MASK = Limbs29<N>::MASK;
  uint32_t x2[L], m[L];
  PBC_COUNT_MACS(L * (L + 1) / 2 + L * L);
#pragma unroll
  for (int i = 0; i < L; i++) x2[i] = x[i] << 1;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc += (uint64_t) x[i] * x2[k - i];
    if ((k & 1) == 0) acc += (uint64_t) x[k / 2] * x[k / 2];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t) m[i] * K.p29[k - i];
    m[k] = ((uint32_t) acc * K.ninv29) & MASK;
    acc += (uint64_t) m[k] * K.p29[0];
    acc >>= Limbs29<N>::W;
  }
#pragma unroll
  for (int k = L; k < 2 * L; k++) {
#pragma unroll
    for (int i = k - L + 1; 2 * i < k; i++) acc += (uint64_t) x[i] * x2[k - i];
    if ((k & 1) == 0) acc += (uint64_t) x[k / 2] * x[k / 2];
#pragma unroll
    for (int i = k - L + 1; i < L; i++) acc += (uint64_t) m[i] * K.p29[k - i];
    t[k - L] = (uint32_t) acc & MASK;
    acc >>= Limbs29<N>::W;
  }
}

// ---------------------------------------------------------------------------------------
// Limb-domain sum of products (lazy reduction for the extension towers):
//     r = (x_0 y_0 + ... + x_{T-1} y_{T-1}) / R  mod q,   one Montgomery reduction for T products.
// Operands and result stay in 29-bit limb form; results are < 2q (normalised limbs) and may be
// fed straight into further sums.  Column bound: (T L + L) 2^58 < 2^64  ->  T <= 9 for L = 6.
// Doubled operands (limbs < 2^30) count as two terms.
// ---------------------------------------------------------------------------------------
template <int N>
struct fl {
  uint32_t l[Limbs29<N>::L];
#ifdef PBC_HOSTSIM
  double hs_u = 1.0, hs_B = 1.0;   // worst-case limb size (units of 2^29) and value (units of q): tracked and asserted by
#endif                             // the redundant-representation code of pairing_al.cuh in the host mirror only
};
template <int N>
PBC_DEV void to_limbs(fl<N> &r, const fp<N> &a) { to29<N>(r.l, a); }
template <int N>
PBC_DEV void limbs_dbl(fl<N> &r, const fl<N> &a) {
#pragma unroll
  for (int i = 0; i < Limbs29<N>::L; i++) r.l[i] = a.l[i] << 1;
}
// value < 2q in limb form -> fully reduced words
template <int N>
PBC_DEV void from_limbs(fp<N> &r, const fl<N> &a) {
  uint32_t w[N];
  uint32_t carry = from29<N>(w, a.l);
  fp_cond_sub<N>(r, w, carry);
}
// keeps the compiler from re-associating separate accumulator chains back into one
#ifdef PBC_HOSTSIM
#define PBC_OPAQUE64(x) ((void) 0)
// host mirror only: the exact column sums of a sum of products in 128 bits -- none may reach 2^64
template <int N, class X>
static inline void hs_sop_check(const X &x, const X &y, int T) {
  constexpr int L = Limbs29<N>::L;
  const FpK<N> &K = fpk<N>();
  unsigned __int128 acc = 0;
  uint32_t m[L];
  for (int k = 0; k < 2 * L; k++) {
    for (int t = 0; t < T; t++)
      for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); i++) acc += (unsigned __int128) x[t].l[i] * y[t].l[k - i];
    for (int i = (k < L ? 0 : k - L + 1); i < (k < L ? k : L); i++) acc += (unsigned __int128) m[i] * K.p29[k - i];
    if (k < L) {
      m[k] = ((uint32_t) acc * K.ninv29) & Limbs29<N>::MASK;
      acc += (unsigned __int128) m[k] * K.p29[0];
    }
    if (acc >> 64) { fprintf(stderr, "hostsim: column %d of a sum of %d products overflows 64 bits\n", k, T); abort(); }
    acc >>= Limbs29<N>::W;
  }
  if (acc) { fprintf(stderr, "hostsim: a Montgomery product does not fit its limbs\n"); abort(); }
}
#define PBC_HS_SOP_CHECK(x, y, T) hs_sop_check<N>(x, y, T)
#else
#define PBC_OPAQUE64(x) asm("" : "+v"(x))
#define PBC_HS_SOP_CHECK(x, y, T) ((void) 0)
#endif
#ifndef PBC_SOP_CHAINS
#define PBC_SOP_CHAINS 1    // independent accumulator chains per column (experiment: the multiply-add chain of a column is
#endif                      // serially dependent; more chains = more instruction-level parallelism, a few extra 64-bit adds)
// (Round 6: the compiler keeps the products of a column in a chain of their own next to the carry chain -- one 64-bit
// addition per column to join them, ~10 of the ~110 instructions of a one-product sum -- however the source orders the
// additions.  Forcing ONE chain with opaque barriers removes those additions (cyc_pair: 1856 -> 1753 vector
// instructions) but back-to-back dependent v_mad_u64_u32 need an s_nop each (937 in cyc_pair), and the kernels ran SLOWER:
// f.param 22.51 -> 22.94 ms, d159 products 157.8 -> 166.4 ms, single d159 pairings unchanged -- profiles/r06_notes.md.)
template <int N, int T, int DBL = 0>      // DBL: how many of the T terms have a doubled operand
PBC_DEV void sop_limbs(fl<N> &r, const fl<N> (&x)[T], const fl<N> (&y)[T]) {
  const FpK<N> &K = fpk<N>();
  constexpr int L = Limbs29<N>::L;
  constexpr uint32_t MASK = Limbs29<N>::MASK;
  static_assert(Limbs29<N>::W == 29 && (T + DBL) * L + L <= 63, "column accumulator would overflow");
  PBC_COUNT_MACS((T + 1) * L * L);
  constexpr int C = PBC_SOP_CHAINS < T ? PBC_SOP_CHAINS : T;
  uint32_t m[L];
  uint64_t acc = 0;
  PBC_HS_SOP_CHECK(x, y, T);
#pragma unroll
  for (int k = 0; k < L; k++) {
    uint64_t part[C];
#pragma unroll
    for (int c = 0; c < C; c++) part[c] = 0;
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int i = 0; i <= k; i++) part[t % C] += (uint64_t) x[t].l[i] * y[t].l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t) m[i] * K.p29[k - i];
#pragma unroll
    for (int c = 0; c < C; c++) { if (C > 1) PBC_OPAQUE64(part[c]); acc += part[c]; }
    m[k] = ((uint32_t) acc * K.ninv29) & MASK;
    acc += (uint64_t) m[k] * K.p29[0];
    acc >>= Limbs29<N>::W;
  }
#pragma unroll
  for (int k = L; k < 2 * L; k++) {
    uint64_t part[C];
#pragma unroll
    for (int c = 0; c < C; c++) part[c] = 0;
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int i = k - L + 1; i < L; i++) part[t % C] += (uint64_t) x[t].l[i] * y[t].l[k - i];
#pragma unroll
    for (int i = k - L + 1; i < L; i++) acc += (uint64_t) m[i] * K.p29[k - i];
#pragma unroll
    for (int c = 0; c < C; c++) { if (C > 1) PBC_OPAQUE64(part[c]); acc += part[c]; }
    r.l[k - L] = (uint32_t) acc & MASK;
    acc >>= Limbs29<N>::W;
  }
}

// Wide accumulator form of the same idea for rolled loops with a run-time number of terms:
// 2L 64-bit column accumulators take products without any carry handling; one Montgomery
// reduction at the end.  Capacity: `units` <= 9 for L = 6 (a term with a doubled operand
// counts as two units): (units + 1) L 2^58 < 2^64.
template <int N>
struct wide {
  uint64_t c[2 * Limbs29<N>::L - 1];
};
template <int N>
PBC_DEV void wide_zero(wide<N> &W) {
#pragma unroll
  for (int k = 0; k < 2 * Limbs29<N>::L - 1; k++) W.c[k] = 0;
}
template <int N>
PBC_DEV void wide_mac(wide<N> &W, const fl<N> &x, const fl<N> &y) {
  constexpr int L = Limbs29<N>::L;
  PBC_COUNT_MACS(L * L);
#pragma unroll
  for (int i = 0; i < L; i++)
#pragma unroll
    for (int j = 0; j < L; j++) {
#ifdef PBC_HOSTSIM
      if (__builtin_add_overflow(W.c[i + j], (uint64_t) x.l[i] * y.l[j], &W.c[i + j])) { fprintf(stderr, "hostsim: wide accumulator column %d overflows\n", i + j); abort(); }
#else
      W.c[i + j] += (uint64_t) x.l[i] * y.l[j];
#endif
    }
}
// Relief of a wide accumulator WITHOUT a reduction: the bits of column k above 2^58 move to column k + 2 (the same
// weight: 2^(29 k) 2^58 = 2^(29 (k + 2))), in ascending order, so that bits moved up are cut again further on.  Exact
// integer bookkeeping -- the value of the accumulator does not change.  Afterwards every column k <= 2L - 4 is below
// 2^58, i.e. 1/L of a product unit: callers carry on with units = 1.  Columns that take fewer than four products per unit
// (the three at either end) are not cut: 3 (kWideMaxUnits + 1) 2^58 < 2^64 (the callers' static accounting keeps a sum
// within kWideMaxUnits).
// Four simple instructions per column against the 2 L^2 multiply-adds of a reduction and a product by R mod q.
constexpr int kWideMaxUnits = 16;
#ifndef PBC_SQZ_NARROW
#define PBC_SQZ_NARROW 1
#endif
template <int N>
PBC_DEV void wide_squeeze(wide<N> &W) {
  constexpr int L = Limbs29<N>::L;
#pragma unroll
  for (int k = 0; k + 2 < 2 * L - 1; k++) {
    // a column that takes p products per unit holds 64 / p units: with at most kWideMaxUnits in a sum only p >= 4 can fill
    if (PBC_SQZ_NARROW && (k + 1 < 4 || 2 * L - 1 - k < 4)) continue;
    const uint64_t hi = W.c[k] >> 58;
    W.c[k] &= (1ull << 58) - 1;
    W.c[k + 2] += hi;
  }
}
template <int N>
PBC_DEV void wide_reduce(fl<N> &r, const wide<N> &W) {
  const FpK<N> &K = fpk<N>();
  constexpr int L = Limbs29<N>::L;
  constexpr uint32_t MASK = Limbs29<N>::MASK;
  PBC_COUNT_MACS(L * L);
  uint32_t m[L];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
    acc += W.c[k];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t) m[i] * K.p29[k - i];
    m[k] = ((uint32_t) acc * K.ninv29) & MASK;
    acc += (uint64_t) m[k] * K.p29[0];
    acc >>= Limbs29<N>::W;
  }
#pragma unroll
  for (int k = L; k < 2 * L; k++) {
    if (k < 2 * L - 1) acc += W.c[k];
#pragma unroll
    for (int i = k - L + 1; i < L; i++) acc += (uint64_t) m[i] * K.p29[k - i];
    r.l[k - L] = (uint32_t) acc & MASK;
    acc >>= Limbs29<N>::W;
  }
}

#ifndef PBC_MUL_IMPL
#define PBC_MUL_IMPL 1      // 1: one column accumulator; 2: two (shorter dependent chains; measured equal at 2 waves per SIMD)
#endif
template <int N>
PBC_DEV void fp_mul_inl(fp<N> &r, const fp<N> &a, const fp<N> &b) {
  fp_mul29_inl<N, PBC_MUL_IMPL == 2>(r, a, b);
}
template <int N>
PBC_DEV void fp_sqr_inl(fp<N> &r, const fp<N> &a) {
  fp_sqr29_inl<N, PBC_MUL_IMPL == 2>(r, a);
}

// Out-of-line instances: one copy of each ~900-instruction body per kernel keeps the Miller
// loop inside the instruction cache.  Operands travel as ext_vector values: clang's AMDGPU
// ABI passes only 16 registers' worth of *aggregates* directly (the second fp<16> struct
// went through scratch memory, 128 B/lane per call), vectors are always passed in VGPRs.
template <int N>
struct vecN {
  typedef uint32_t type __attribute__((ext_vector_type(N)));
};
template <int N>
PBC_DEV typename vecN<N>::type to_vec(const fp<N> &a) {
  typename vecN<N>::type v;
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = a.v[i];
  return v;
}
template <int N>
PBC_DEV void from_vec(fp<N> &r, typename vecN<N>::type v) {
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = v[i];
}
template <int N>
static __device__ __noinline__ typename vecN<N>::type fp_mul_fn(typename vecN<N>::type va, typename vecN<N>::type vb) {
  fp<N> a, b, r;
  from_vec<N>(a, va);
  from_vec<N>(b, vb);
  fp_mul_inl<N>(r, a, b);
  return to_vec<N>(r);
}
#ifndef PBC_INLINE_MUL
#define PBC_INLINE_MUL 0    // 1: every product inlined at its call site (experiment)
#endif
#ifndef PBC_INLINE_SMALL
#define PBC_INLINE_SMALL 0  // 1: products of the 160-bit fields are inlined (a call costs more than the
#endif                      //    78 multiply-adds it wraps); code size is held by the tower-level calls
// Wide fields (33 words: types a1, e): an element no longer fits the argument registers, and moving
// 99 words through them at every call site bloats the loops past the instruction cache.  Operands
// and results of the out-of-line routines are passed by address instead: the elements live in the
// lane's private memory (scratch, dword-interleaved across the wave, so accesses are coalesced) and a
// call site is a handful of scalar instructions.
#ifndef PBC_MEM_OPERANDS
#define PBC_MEM_OPERANDS 1
#endif
template <int N> constexpr bool kMemOperands = PBC_MEM_OPERANDS && N >= 32;
template <int N>
static __device__ __noinline__ void fp_mul_mem(fp<N> *r, const fp<N> *a, const fp<N> *b) {
  fp<N> x = *a, y = *b, z;
  fp_mul_inl<N>(z, x, y);
  *r = z;
}
template <int N>
static __device__ __noinline__ void fp_sqr_mem(fp<N> *r, const fp<N> *a) {
  fp<N> x = *a, z;
  fp_sqr_inl<N>(z, x);
  *r = z;
}
template <int N>
PBC_DEV void fp_mul(fp<N> &r, const fp<N> &a, const fp<N> &b) {
#if PBC_INLINE_MUL
  fp_mul_inl<N>(r, a, b);
#else
  if constexpr (kMemOperands<N>) {
    fp_mul_mem<N>(&r, &a, &b);
    return;
  }
  if constexpr (N <= 5 && PBC_INLINE_SMALL) {
    fp_mul_inl<N>(r, a, b);
    return;
  }
  // 16-word operands need 32 argument VGPRs and the ABI has 31: the last word of b goes through
  // the stack.  Routing it through LDS instead was measured and changed nothing (profiles/r01_notes.md).
  from_vec<N>(r, fp_mul_fn<N>(to_vec<N>(a), to_vec<N>(b)));
#endif
}
// The reference has no dedicated Fq squaring (generic_square = mul(a,a), arith/field.c:383);
// here it is its own out-of-line body with ~3/4 of the multiply-adds.
template <int N>
static __device__ __noinline__ typename vecN<N>::type fp_sqr_fn(typename vecN<N>::type va) {
  fp<N> a, r;
  from_vec<N>(a, va);
  fp_sqr_inl<N>(r, a);
  return to_vec<N>(r);
}
template <int N>
PBC_DEV void fp_sqr(fp<N> &r, const fp<N> &a) {
#if PBC_INLINE_MUL
  fp_sqr_inl<N>(r, a);
#else
  if constexpr (N <= 5 && PBC_INLINE_SMALL) {
    fp_sqr_inl<N>(r, a);
    return;
  }
  if constexpr (kMemOperands<N>) {
    fp_sqr_mem<N>(&r, &a);
    return;
  }
  from_vec<N>(r, fp_sqr_fn<N>(to_vec<N>(a)));
#endif
}

// fp_add (montfp.c:220-250)
template <int N>
PBC_DEV void fp_add_inl(fp<N> &r, const fp<N> &a, const fp<N> &b) {
  uint32_t t[N];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  fp_cond_sub<N>(r, t, c);
}
// fp_double (montfp.c:252-270)
template <int N>
PBC_DEV void fp_dbl_inl(fp<N> &r, const fp<N> &a) {
  uint32_t t[N];
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = i ? __builtin_amdgcn_alignbit(a.v[i], a.v[i - 1], 31) : a.v[0] << 1;
  fp_cond_sub<N>(r, t, a.v[N - 1] >> 31);
}
// fp_sub (montfp.c:282-316)
template <int N>
PBC_DEV void fp_sub_inl(fp<N> &r, const fp<N> &a, const fp<N> &b) {
  const FpK<N> &K = fpk<N>();
  uint32_t d[N];
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < N; i++) d[i] = __builtin_subc(a.v[i], b.v[i], bw, &bw);
  uint32_t mask = 0u - bw;      // add q back when the subtraction borrowed
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = __builtin_addc(d[i], K.p[i] & mask, c, &c);
}
// add / double / subtract: inline for the register-resident fields, out of line on memory operands
// for the wide ones (kMemOperands)
template <int N>
static __device__ __noinline__ void fp_add_mem(fp<N> *r, const fp<N> *a, const fp<N> *b) {
  fp<N> x = *a, y = *b, z;
  fp_add_inl<N>(z, x, y);
  *r = z;
}
template <int N>
static __device__ __noinline__ void fp_sub_mem(fp<N> *r, const fp<N> *a, const fp<N> *b) {
  fp<N> x = *a, y = *b, z;
  fp_sub_inl<N>(z, x, y);
  *r = z;
}
template <int N>
static __device__ __noinline__ void fp_dbl_mem(fp<N> *r, const fp<N> *a) {
  fp<N> x = *a, z;
  fp_dbl_inl<N>(z, x);
  *r = z;
}
template <int N>
PBC_DEV void fp_add(fp<N> &r, const fp<N> &a, const fp<N> &b) {
  if constexpr (kMemOperands<N>) fp_add_mem<N>(&r, &a, &b);
  else fp_add_inl<N>(r, a, b);
}
template <int N>
PBC_DEV void fp_sub(fp<N> &r, const fp<N> &a, const fp<N> &b) {
  if constexpr (kMemOperands<N>) fp_sub_mem<N>(&r, &a, &b);
  else fp_sub_inl<N>(r, a, b);
}
template <int N>
PBC_DEV void fp_dbl(fp<N> &r, const fp<N> &a) {
  if constexpr (kMemOperands<N>) fp_dbl_mem<N>(&r, &a);
  else fp_dbl_inl<N>(r, a);
}
// ---------------------------------------------------------------------------------------
// Fused products (round 5; the wide fields of types a1 / e).  On memory operands every sum, difference and doubling
// of a step was its own call that loads 33-66 words and stores 33 (half of a step's traffic through private memory --
// which reaches HBM: 2.1 MB per type e pairing, profiles/r05_ab_wide.txt -- and a load-to-use stall each).  One call
//     r = 2^d ( (a [+- a2]) (k b [+- b2]) [+- 2^s1 c1] [+- 2^s2 c2] )
// takes the linear work into the product's registers; every intermediate is the canonical representative mod q, so a
// fused step returns what the same step built from fp_add / fp_sub / fp_mul returns.  `op` (wave-uniform) selects:
template <int N> PBC_DEV void fp_neg_inl(fp<N> &r, const fp<N> &a);   // (below)
#ifndef PBC_WIDE_NO_SOP
#define PBC_WIDE_NO_SOP 0              // 1: the steps of types a1 / e without the two-product sums fp_sopx (same-box A/B)
#endif
namespace fx {
constexpr int A_ADD = 1, A_SUB = 2, B_ADD = 4, B_SUB = 8, C1_ADD = 16, C1_SUB = 32, C2_ADD = 256, C2_SUB = 512;
constexpr int c1_sh(int s) { return s << 6; }        // c1 enters as 2^s c1, s = 0..3
constexpr int c2_sh(int s) { return s << 10; }
constexpr int dbl(int d) { return d << 12; }         // the result is doubled d times, d = 0..3
constexpr int b_times(int k) { return k << 16; }     // b enters as k b, k = 2..255 (0, 1: b itself)
constexpr int NEG2 = 1 << 14;                        // fp_sopx: the second product is subtracted
}
template <int N>
PBC_DEV void fx_addsub(fp<N> &x, const fp<N> *p, int mode, int sh) {
  fp<N> t = *p;
  for (int i = 0; i < sh; i++) fp_dbl_inl<N>(t, t);
  if (mode & 1) fp_add_inl<N>(x, x, t);
  else fp_sub_inl<N>(x, x, t);
}
template <int N>
PBC_DEV void fx_finish(fp<N> *r, fp<N> &z, const fp<N> *c1, const fp<N> *c2, int op) {
  if (op & 48) fx_addsub<N>(z, c1, (op >> 4) & 3, (op >> 6) & 3);
  if (op & 768) fx_addsub<N>(z, c2, (op >> 8) & 3, (op >> 10) & 3);
  for (int i = 0; i < ((op >> 12) & 3); i++) fp_dbl_inl<N>(z, z);
  *r = z;
}
template <int N>
PBC_DEV void fx_mul_body(fp<N> *r, const fp<N> *a, const fp<N> *a2, const fp<N> *b, const fp<N> *b2, const fp<N> *c1,
                         const fp<N> *c2, int op) {
  fp<N> x = *a, y = *b, z;
  if (op & 3) fx_addsub<N>(x, a2, op & 3, 0);
  const int k = (op >> 16) & 255;
  if (k > 1) {                                       // k b by double-and-add (k is a small constant of the parameter set)
    const fp<N> base = y;
    for (int i = 30 - __builtin_clz(k); i >= 0; i--) {
      fp_dbl_inl<N>(y, y);
      if ((k >> i) & 1) fp_add_inl<N>(y, y, base);
    }
  }
  if (op & 12) fx_addsub<N>(y, b2, (op >> 2) & 3, 0);
  fp_mul_inl<N>(z, x, y);
  fx_finish<N>(r, z, c1, c2, op);
}
template <int N>
PBC_DEV void fx_sqr_body(fp<N> *r, const fp<N> *a, const fp<N> *a2, const fp<N> *c1, const fp<N> *c2, int op) {
  fp<N> x = *a, z;
  if (op & 3) fx_addsub<N>(x, a2, op & 3, 0);
  fp_sqr_inl<N>(z, x);
  fx_finish<N>(r, z, c1, c2, op);
}
template <int N>
static __device__ __noinline__ void fp_mulx_mem(fp<N> *r, const fp<N> *a, const fp<N> *a2, const fp<N> *b, const fp<N> *b2,
                                                const fp<N> *c1, const fp<N> *c2, int op_) {
#ifdef PBC_HOSTSIM
  const int op = op_;
#else
  const int op = __builtin_amdgcn_readfirstlane(op_);     // arguments arrive in vector registers; the selector is uniform
#endif
  fx_mul_body<N>(r, a, a2, b, b2, c1, c2, op);
}
template <int N>
static __device__ __noinline__ void fp_sqrx_mem(fp<N> *r, const fp<N> *a, const fp<N> *a2, const fp<N> *c1, const fp<N> *c2, int op_) {
#ifdef PBC_HOSTSIM
  const int op = op_;
#else
  const int op = __builtin_amdgcn_readfirstlane(op_);
#endif
  fx_sqr_body<N>(r, a, a2, c1, c2, op);
}
// r = 2^dd ( a b +- c (k d [+- d2]) [+- 2^s1 c1] [+- 2^s2 c2] ) with one reduction for the two products (fx::NEG2 selects
// the minus; B_ADD / B_SUB / b_times apply to d)
template <int N>
static __device__ __noinline__ void fp_sopx_mem(fp<N> *r, const fp<N> *a, const fp<N> *b, const fp<N> *c, const fp<N> *d, const fp<N> *d2,
                                                const fp<N> *c1, const fp<N> *c2, int op_) {
#ifdef PBC_HOSTSIM
  const int op = op_;
#else
  const int op = __builtin_amdgcn_readfirstlane(op_);
#endif
  fp<N> x = *a, y = *b, u = *c, v = *d, z;
  const int k = (op >> 16) & 255;
  if (k > 1) {
    const fp<N> base = v;
    for (int i = 30 - __builtin_clz(k); i >= 0; i--) {
      fp_dbl_inl<N>(v, v);
      if ((k >> i) & 1) fp_add_inl<N>(v, v, base);
    }
  }
  if (op & 12) fx_addsub<N>(v, d2, (op >> 2) & 3, 0);
  if (op & fx::NEG2) fp_neg_inl<N>(v, v);
  fp_sop2_29_inl<N, PBC_MUL_IMPL == 2>(z, x, y, u, v);
  fx_finish<N>(r, z, c1, c2, op);
}
template <int N>
PBC_DEV void fp_sopx(fp<N> &r, int op, const fp<N> &a, const fp<N> &b, const fp<N> &c, const fp<N> &d, const fp<N> &d2,
                     const fp<N> &c1, const fp<N> &c2) {
  if constexpr (kMemOperands<N>) {
    fp_sopx_mem<N>(&r, &a, &b, &c, &d, &d2, &c1, &c2, op);
  } else {                             // register-resident fields: two products of their out-of-line body
    fp<N> v = d, z, t;
    const int k = (op >> 16) & 255;
    if (k > 1) {
      const fp<N> base = v;
      for (int i = 30 - __builtin_clz(k); i >= 0; i--) {
        fp_dbl_inl<N>(v, v);
        if ((k >> i) & 1) fp_add_inl<N>(v, v, base);
      }
    }
    if (op & 12) fx_addsub<N>(v, &d2, (op >> 2) & 3, 0);
    fp_mul<N>(t, c, v);
    fp_mul<N>(z, a, b);
    if (op & fx::NEG2) fp_sub_inl<N>(z, z, t);
    else fp_add_inl<N>(z, z, t);
    fx_finish<N>(&r, z, &c1, &c2, op);
  }
}
// r may be any of the operands (everything is read before r is written).  Fields whose elements live in registers run
// the same sequence on the out-of-line product of their width.
template <int N>
PBC_DEV void fp_mulx(fp<N> &r, int op, const fp<N> &a, const fp<N> &a2, const fp<N> &b, const fp<N> &b2, const fp<N> &c1,
                     const fp<N> &c2) {
  if constexpr (kMemOperands<N>) {
    fp_mulx_mem<N>(&r, &a, &a2, &b, &b2, &c1, &c2, op);
  } else {
    fp<N> x = a, y = b, z;
    if (op & 3) fx_addsub<N>(x, &a2, op & 3, 0);
    const int k = (op >> 16) & 255;
    if (k > 1) {
      const fp<N> base = y;
      for (int i = 30 - __builtin_clz(k); i >= 0; i--) {
        fp_dbl_inl<N>(y, y);
        if ((k >> i) & 1) fp_add_inl<N>(y, y, base);
      }
    }
    if (op & 12) fx_addsub<N>(y, &b2, (op >> 2) & 3, 0);
    fp_mul<N>(z, x, y);
    fx_finish<N>(&r, z, &c1, &c2, op);
  }
}
template <int N>
PBC_DEV void fp_sqrx(fp<N> &r, int op, const fp<N> &a, const fp<N> &a2, const fp<N> &c1, const fp<N> &c2) {
  if constexpr (kMemOperands<N>) {
    fp_sqrx_mem<N>(&r, &a, &a2, &c1, &c2, op);
  } else {
    fp<N> x = a, z;
    if (op & 3) fx_addsub<N>(x, &a2, op & 3, 0);
    fp_sqr<N>(z, x);
    fx_finish<N>(&r, z, &c1, &c2, op);
  }
}
// shorthands: operands that an `op` does not name are never read
template <int N> PBC_DEV void fp_mulx(fp<N> &r, const fp<N> &a, const fp<N> &b) { fp_mulx<N>(r, 0, a, a, b, b, a, a); }
template <int N> PBC_DEV void fp_sqrx(fp<N> &r, const fp<N> &a) { fp_sqrx<N>(r, 0, a, a, a, a); }

template <int N>
PBC_DEV bool fp_is0(const fp<N> &a) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < N; i++) x |= a.v[i];
  return x == 0;
}
template <int N>
PBC_DEV bool fp_eq(const fp<N> &a, const fp<N> &b) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < N; i++) x |= a.v[i] ^ b.v[i];
  return x == 0;
}
// fp_neg (montfp.c:318-330): 0 stays 0
template <int N>
PBC_DEV void fp_neg_inl(fp<N> &r, const fp<N> &a) {
  const FpK<N> &K = fpk<N>();
  uint32_t mask = fp_is0<N>(a) ? 0u : 0xffffffffu;
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = __builtin_subc(K.p[i] & mask, a.v[i], bw, &bw);
}
// fp_halve (montfp.c:272-280): a/2 = (a + (a odd ? q : 0)) >> 1
template <int N>
PBC_DEV void fp_halve_inl(fp<N> &r, const fp<N> &a) {
  const FpK<N> &K = fpk<N>();
  uint32_t mask = 0u - (a.v[0] & 1);
  uint32_t t[N];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = __builtin_addc(a.v[i], K.p[i] & mask, c, &c);
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = __builtin_amdgcn_alignbit(i + 1 < N ? t[i + 1] : c, t[i], 1);
}
template <int N>
PBC_DEV void fp_set(fp<N> &r, const uint32_t *w) {
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = w[i];
}
template <int N>
PBC_DEV void fp_cmov_inl(fp<N> &r, const fp<N> &a, bool take) {
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = take ? a.v[i] : r.v[i];
}
template <int N>
static __device__ __noinline__ void fp_neg_mem(fp<N> *r, const fp<N> *a) {
  fp<N> x = *a, z;
  fp_neg_inl<N>(z, x);
  *r = z;
}
template <int N>
static __device__ __noinline__ void fp_halve_mem(fp<N> *r, const fp<N> *a) {
  fp<N> x = *a, z;
  fp_halve_inl<N>(z, x);
  *r = z;
}
template <int N>
static __device__ __noinline__ void fp_cmov_mem(fp<N> *r, const fp<N> *a, bool take) {
  fp<N> x = *a, z = *r;
  fp_cmov_inl<N>(z, x, take);
  *r = z;
}
template <int N>
PBC_DEV void fp_neg(fp<N> &r, const fp<N> &a) {
  if constexpr (kMemOperands<N>) fp_neg_mem<N>(&r, &a);
  else fp_neg_inl<N>(r, a);
}
template <int N>
PBC_DEV void fp_halve(fp<N> &r, const fp<N> &a) {
  if constexpr (kMemOperands<N>) fp_halve_mem<N>(&r, &a);
  else fp_halve_inl<N>(r, a);
}
template <int N>
PBC_DEV void fp_cmov(fp<N> &r, const fp<N> &a, bool take) {
  if constexpr (kMemOperands<N>) fp_cmov_mem<N>(&r, &a, take);
  else fp_cmov_inl<N>(r, a, take);
}

// ---------------------------------------------------------------------------------------
// Inversion: Bernstein-Yang "safegcd" divsteps, constant time, wave-uniform control flow.
// fp_invert in the reference (montfp.c:401-422) calls mpz_invert and multiplies by R^3; the
// inverse is unique, so the residue is identical.  0 -> 0.
//
// A Fermat ladder costs ~1.5 bits(q) F_q products (14 % of a Type-A pairing); divsteps needs
// ~1180 cheap scalar steps for a 512-bit modulus: batches of 30 steps on the low words build a
// 2x2 transition matrix that is then applied to the full-width (f, g) and, modulo q, (d, e),
// all in signed 30-bit limbs (the structure of libsecp256k1's modinv32, re-derived here for
// arbitrary limb counts).  Iteration bound for the half-delta variant:
// floor((45907 b + 26313) / 19929) divsteps for b-bit moduli (b >= 46): 1180 for 512 bits
// (40 batches), 370 for 160 bits (13 batches).
// ---------------------------------------------------------------------------------------
template <int N>
struct Inv30 {
  static constexpr int L = (32 * N + 2 + 29) / 30;                      // sign + one spare bit
  static constexpr int STEPS = (45907 * (32 * N) + 26313) / 19929;
  static constexpr int BATCHES = (STEPS + 29) / 30;
  static constexpr int32_t M30 = (int32_t) (0xffffffffu >> 2);
};

// 30 divsteps on the low words; t = [[u v],[q r]] with t * [f, g] = 2^30 * [f', g']
PBC_DEV int32_t inv_divsteps30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t &tu, int32_t &tv,
                               int32_t &tq, int32_t &tr) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 2
  for (int i = 0; i < 30; i++) {
    uint32_t c1 = (uint32_t) (zeta >> 31);      // zeta < 0
    uint32_t c2 = 0u - (g & 1);                 // g odd
    uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
    g += x & c2;
    q += y & c2;
    r += z & c2;
    c1 &= c2;
    zeta = (int32_t) ((uint32_t) zeta ^ c1) - 1;
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  tu = (int32_t) u; tv = (int32_t) v; tq = (int32_t) q; tr = (int32_t) r;
  return zeta;
}

template <int N>
static __device__ __noinline__ typename vecN<N>::type fp_inv_fn(typename vecN<N>::type va) {
  const FpK<N> &K = fpk<N>();
  constexpr int L = Inv30<N>::L;
  constexpr int32_t M30 = Inv30<N>::M30;
  int32_t f[L], g[L], d[L], e[L];
  // f = q, g = a (the Montgomery residue, as an integer), d = 0, e = 1
#pragma unroll
  for (int i = 0; i < L; i++) {
    const int bit = 30 * i, j = bit >> 5, sh = bit & 31;
    uint32_t x = 0;
    if (j < N) {
      x = va[j] >> sh;
      if (sh > 2 && j + 1 < N) x |= va[j + 1] << (32 - sh);
    }
    g[i] = (int32_t) (x & (uint32_t) M30);
    f[i] = (int32_t) K.p30[i];
    d[i] = 0;
    e[i] = (i == 0);
  }
  int32_t zeta = -1;
  PBC_COUNT_MACS((10 * L + 2) * Inv30<N>::BATCHES);      // signed 32 x 32 + 64 multiply-adds of the matrix applications
#pragma nounroll
  for (int it = 0; it < Inv30<N>::BATCHES; it++) {
    int32_t u, v, q, r;
    zeta = inv_divsteps30(zeta, (uint32_t) f[0], (uint32_t) g[0], u, v, q, r);
    // (d, e) <- t (d, e) / 2^30 mod q, kept in (-2q, q)
    {
      int32_t sd = d[L - 1] >> 31, se = e[L - 1] >> 31;
      int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
      int64_t cd = (int64_t) u * d[0] + (int64_t) v * e[0];
      int64_t ce = (int64_t) q * d[0] + (int64_t) r * e[0];
      md -= (int32_t) ((K.qinv30 * (uint32_t) cd + (uint32_t) md) & (uint32_t) M30);
      me -= (int32_t) ((K.qinv30 * (uint32_t) ce + (uint32_t) me) & (uint32_t) M30);
      cd += (int64_t) (int32_t) K.p30[0] * md;
      ce += (int64_t) (int32_t) K.p30[0] * me;
      cd >>= 30;
      ce >>= 30;
#pragma unroll
      for (int i = 1; i < L; i++) {
        cd += (int64_t) u * d[i] + (int64_t) v * e[i];
        ce += (int64_t) q * d[i] + (int64_t) r * e[i];
        cd += (int64_t) (int32_t) K.p30[i] * md;
        ce += (int64_t) (int32_t) K.p30[i] * me;
        d[i - 1] = (int32_t) cd & M30;
        e[i - 1] = (int32_t) ce & M30;
        cd >>= 30;
        ce >>= 30;
      }
      d[L - 1] = (int32_t) cd;
      e[L - 1] = (int32_t) ce;
    }
    // (f, g) <- t (f, g) / 2^30 (exact)
    {
      int64_t cf = (int64_t) u * f[0] + (int64_t) v * g[0];
      int64_t cg = (int64_t) q * f[0] + (int64_t) r * g[0];
      cf >>= 30;
      cg >>= 30;
#pragma unroll
      for (int i = 1; i < L; i++) {
        cf += (int64_t) u * f[i] + (int64_t) v * g[i];
        cg += (int64_t) q * f[i] + (int64_t) r * g[i];
        f[i - 1] = (int32_t) cf & M30;
        g[i - 1] = (int32_t) cg & M30;
        cf >>= 30;
        cg >>= 30;
      }
      f[L - 1] = (int32_t) cf;
      g[L - 1] = (int32_t) cg;
    }
  }
  // now g = 0, f = +-1 and d = +-a^-1: normalise d into [0, q), negating when f = -1
  {
    int32_t sign = f[L - 1] >> 31;
    int32_t cond_add = d[L - 1] >> 31;
#pragma unroll
    for (int i = 0; i < L; i++) {
      d[i] += (int32_t) K.p30[i] & cond_add;
      d[i] = (d[i] ^ sign) - sign;
    }
#pragma unroll
    for (int i = 0; i + 1 < L; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    cond_add = d[L - 1] >> 31;
#pragma unroll
    for (int i = 0; i < L; i++) d[i] += (int32_t) K.p30[i] & cond_add;
#pragma unroll
    for (int i = 0; i + 1 < L; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
  }
  // 30-bit limbs -> words; the value is (a R)^-1, times R^3 / R gives a^-1 R
  fp<N> t, r3, res;
#pragma unroll
  for (int j = 0; j < N; j++) {
    const int bit = 32 * j, i = bit / 30, o = bit - 30 * i;
    uint32_t x = (uint32_t) d[i] >> o;
    if (i + 1 < L) x |= (uint32_t) d[i + 1] << (30 - o);
    if (60 - o < 32 && i + 2 < L) x |= (uint32_t) d[i + 2] << (60 - o);
    t.v[j] = x;
  }
  // a == 0: divsteps leaves f = q, d = 0 -> t = 0 (or q after the sign fix-up); force 0
  bool zero = true;
#pragma unroll
  for (int j = 0; j < N; j++) zero &= (va[j] == 0);
  if (zero) {
#pragma unroll
    for (int j = 0; j < N; j++) t.v[j] = 0;
  }
  fp_set<N>(r3, K.r3);
  fp_mul<N>(res, t, r3);
  return to_vec<N>(res);
}
template <int N>
static __device__ __noinline__ void fp_inv_mem(fp<N> *r, const fp<N> *a) {
  fp<N> x = *a, z;
  from_vec<N>(z, fp_inv_fn<N>(to_vec<N>(x)));
  *r = z;
}
template <int N>
PBC_DEV void fp_inv(fp<N> &r, const fp<N> &a) {
  if constexpr (kMemOperands<N>) fp_inv_mem<N>(&r, &a);
  else from_vec<N>(r, fp_inv_fn<N>(to_vec<N>(a)));
}

// bytes per F_q coordinate (fixed_length_in_bytes of the field)
template <int N>
PBC_DEV int fq_bytes() { return (int) fpk<N>().fbytes; }

// Wire format: fixed-width big-endian canonical residue (fp_from_bytes montfp.c:498-517,
// fp_to_bytes :487-496 + pbc_mpz_out_raw_n field.c:629-638).  The byte length is
// ceil(bits(q)/8): a whole number of words for a.param (64) and d159/f (20), not for e.g. the
// 175-, 196- and 201-bit MNT fields (22, 25, 26 bytes).
template <int N>
PBC_DEV void fp_load_be(fp<N> &r, const uint8_t *src) {
  const FpK<N> &K = fpk<N>();
  fp<N> t;
  if (K.fbytes == 4 * N) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(src);
#pragma unroll
    for (int i = 0; i < N; i++) t.v[N - 1 - i] = __builtin_bswap32(w[i]);
  } else {                             // byte lengths that are not whole words (e.g. 175-, 201-bit q)
    const int nb = (int) K.fbytes;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint32_t x = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int pos = nb - 1 - (4 * i + b);
        if (pos >= 0) x |= (uint32_t) src[pos] << (8 * b);
      }
      t.v[i] = x;
    }
  }
  fp<N> r2;
  fp_set<N>(r2, K.r2);
  fp_mul<N>(r, t, r2);            // x -> x R mod q (reduces x >= q as well)
}
template <int N>
PBC_DEV void fp_store_be(uint8_t *dst, const fp<N> &a) {
  const FpK<N> &K = fpk<N>();
  fp<N> one, t;
#pragma unroll
  for (int i = 0; i < N; i++) one.v[i] = (i == 0);
  fp_mul<N>(t, a, one);           // a R^-1: canonical residue
  if (K.fbytes == 4 * N) {
    uint32_t *w = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
    for (int i = 0; i < N; i++) w[i] = __builtin_bswap32(t.v[N - 1 - i]);
  } else {
    const int nb = (int) K.fbytes;
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int pos = nb - 1 - (4 * i + b);
        if (pos >= 0) dst[pos] = (uint8_t) (t.v[i] >> (8 * b));
      }
  }
}

}  // namespace pbc
