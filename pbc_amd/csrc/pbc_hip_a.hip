// pbc_hip_a.hip -- kernels and launches of types a, a1 and e (libpbc_hip.so; see host_common.h)
#include "host_common.h"
#include "pairing_aw.cuh"
#include "pairing_ew.cuh"

// One Type-A pairing per lane.  g1/g2/gt are AoS in wire format (128 B each for a.param);
// per-lane 16-byte loads of a 128-byte record: every byte of every fetched line is used.
// F_q elements are in limb form throughout (pairing_al.cuh).
// lanes per workgroup: 256 for the 33-word kernels (one wave per SIMD of a CU; see launch_a), 128 otherwise
template <int N> constexpr int kWide = N >= 32 ? 256 : kBlock;
// The 33-word kernels below keep 72 words of LDS per lane in 256-lane workgroups: 72 KB of static LDS, which gfx950's
// 160 KB per CU holds twice and the 64 KB of gfx90a / gfx942 not at all -- this library is written for gfx950 only.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libpbc_hip targets gfx950 (MI355X): its kernels are sized for 160 KB of LDS per CU"
#endif
template <int N>
__global__ void __launch_bounds__(kBlock, PBC_A_WAVES) al_pairing_kernel(uint8_t *gt, const uint8_t *g1,
                                                             const uint8_t *g2, size_t n, unsigned *ctr, KArgs<N> ka) {
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;
    constexpr int L = 8 * N;
    __attribute__((aligned(16))) uint8_t out[L];
    AL<N>::pairing_lane(out, g1 + ld * L, g2 + ld * L);
    if (idx < n) {
      uint4 *dst = reinterpret_cast<uint4 *>(gt + idx * L);
      const uint4 *src = reinterpret_cast<const uint4 *>(out);
#pragma unroll
      for (int i = 0; i < L / 16; i++) dst[i] = src[i];
    }
  }
}

// Small batches: one pairing per WAVEFRONT (pairing_aw.cuh: one limb per lane, products across the lanes) -- a third of
// the latency of a lane-local pairing -- or, for the smallest ones, per WORKGROUP of four wavefronts that share the
// independent products of every step (NW = 4).  One workgroup per unit.
#ifndef PBC_AW_WAVES
#define PBC_AW_WAVES 4                 // waves per SIMD the register budget allows (128 VGPRs: the rarely run word-form boundary code and the inversion spill)
#endif
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, PBC_AW_WAVES) aw_pairing_kernel(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, KArgs<N> ka) {
  constexpr int L = 8 * N;
  const size_t idx = blockIdx.x;
  AW<N, NW> w;
  w.pairing_wave(gt + idx * L, g1 + idx * L, g2 + idx * L);
}

// pairing_pp_apply and few-term products on the wave routines (pairing_aw.cuh, round 5): one second argument / one term /
// one product per workgroup of NW wavefronts
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, PBC_AW_WAVES) aw_pp_apply_kernel(uint8_t *gt, const uint32_t *__restrict__ tab, const uint32_t *__restrict__ valid,
                                                                            const uint8_t *g2, KArgs<N> ka) {
  constexpr int L = 8 * N;
  const size_t idx = blockIdx.x;
  AW<N, NW> w;
  w.pp_apply_wave(gt + idx * L, tab, *valid != 0, g2 + idx * L);
}
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, PBC_AW_WAVES) aw_miller_kernel(uint32_t *ws, const uint8_t *g1, const uint8_t *g2, KArgs<N> ka) {
  constexpr int L = 8 * N;
  const size_t idx = blockIdx.x;
  AW<N, NW> w;
  w.miller_record_wave(ws + idx * AW<N, NW>::WREC, g1 + idx * L, g2 + idx * L);
}
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, PBC_AW_WAVES) aw_prod_finish_kernel(uint8_t *gt, const uint32_t *ws, int k, KArgs<N> ka) {
  constexpr int L = 8 * N;
  const size_t idx = blockIdx.x;
  AW<N, NW> w;
  w.prod_finish_wave(gt + idx * L, ws + idx * (size_t) k * AW<N, NW>::WREC, k);
}

// The same four kernels for type a1 and for type a parameters outside the fast path (AW<N, NW, AG<N>>: the Miller loop over the
// signed digits of the order, 38 limbs of 28 bits on the 33-word fields; `aux`: the object's table, ag_aux_build).  One lane
// needs 0.2 s for an a1.param pairing however small the batch.
template <int N> constexpr int kAgWaves = N >= 32 ? 2 : PBC_AW_WAVES;
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, kAgWaves<N>) agw_pairing_kernel(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, const uint32_t *aux, KArgs<N> ka) {
  const size_t L = 2 * (size_t) fq_bytes<N>(), idx = blockIdx.x;
  AW<N, NW, AG<N>> w;
  w.aux = aux;
  w.pairing_wave(gt + idx * L, g1 + idx * L, g2 + idx * L);
}
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, kAgWaves<N>) agw_pp_apply_kernel(uint8_t *gt, const uint32_t *__restrict__ tab, const uint32_t *__restrict__ valid,
                                                                            const uint8_t *g2, const uint32_t *aux, KArgs<N> ka) {
  const size_t L = 2 * (size_t) fq_bytes<N>(), idx = blockIdx.x;
  AW<N, NW, AG<N>> w;
  w.aux = aux;
  w.pp_apply_wave(gt + idx * L, tab, *valid != 0, g2 + idx * L);
}
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, kAgWaves<N>) agw_miller_kernel(uint32_t *ws, const uint8_t *g1, const uint8_t *g2, const uint32_t *aux, KArgs<N> ka) {
  const size_t L = 2 * (size_t) fq_bytes<N>(), idx = blockIdx.x;
  AW<N, NW, AG<N>> w;
  w.aux = aux;
  w.miller_record_wave(ws + idx * AW<N, NW, AG<N>>::WREC, g1 + idx * L, g2 + idx * L);
}
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, kAgWaves<N>) agw_prod_finish_kernel(uint8_t *gt, const uint32_t *ws, int k, const uint32_t *aux, KArgs<N> ka) {
  const size_t L = 2 * (size_t) fq_bytes<N>(), idx = blockIdx.x;
  AW<N, NW, AG<N>> w;
  w.aux = aux;
  w.prod_finish_wave(gt + idx * L, ws + idx * (size_t) k * AW<N, NW, AG<N>>::WREC, k);
}
// ... and type e (pairing_ew.cuh: numerator and denominator, the verticals kept, the (q - 1) / r power on the wave): one lane needs
// 35 ms for an e.param pairing however small the batch
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, kAgWaves<N>) ew_pairing_kernel(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, const uint32_t *aux, KArgs<N> ka) {
  const size_t NB = (size_t) fq_bytes<N>(), idx = blockIdx.x;
  EW<N, NW> w;
  w.aux = aux;
  w.pairing_wave(gt + idx * NB, g1 + idx * 2 * NB, g2 + idx * 2 * NB);
}
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, kAgWaves<N>) ew_miller_kernel(uint32_t *ws, const uint8_t *g1, const uint8_t *g2, const uint32_t *aux, KArgs<N> ka) {
  const size_t NB = (size_t) fq_bytes<N>(), idx = blockIdx.x;
  EW<N, NW> w;
  w.aux = aux;
  w.miller_record_wave(ws + idx * EW<N, NW>::WREC, g1 + idx * 2 * NB, g2 + idx * 2 * NB);
}
template <int N, int NW>
__global__ void __launch_bounds__(64 * NW, kAgWaves<N>) ew_prod_finish_kernel(uint8_t *gt, const uint32_t *ws, int k, const uint32_t *aux, KArgs<N> ka) {
  const size_t NB = (size_t) fq_bytes<N>(), idx = blockIdx.x;
  EW<N, NW> w;
  w.aux = aux;
  w.prod_finish_wave(gt + idx * NB, ws + idx * (size_t) k * EW<N, NW>::WREC, k);
}
// (the table: ~800 bytes, one copy per device the object runs on, uploaded on first use)
static const uint32_t *ag_device_aux(pbc_hip_pairing_s *P) {
  if (P->ag_aux.empty()) return nullptr;
  static const char kAuxKey = 0;
  bool fresh = false;
  uint32_t *d = (uint32_t *) object_scratch(P, &kAuxKey, P->ag_aux.size() * sizeof(uint32_t), &fresh);
  if (!d) return nullptr;
  if (fresh && hipMemcpy(d, P->ag_aux.data(), P->ag_aux.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}
extern "C" size_t pbc_hip_diag_ag_table(pbc_hip_pairing_t *P, uint32_t *out, size_t cap) {
  if (!P) return 0;
  for (size_t i = 0; i < P->ag_aux.size() && i < cap; i++) out[i] = P->ag_aux[i];
  return P->ag_aux.size();
}
static bool ag_capable(const pbc_hip_pairing_s *P) { return (P->type == '1' || (P->type == 'a' && P->a_generic)) && !P->ag_aux.empty(); }
#define PBC_DISPATCH_AG(P, ...) do { if ((P)->nlimb == 16) { constexpr int N = 16; __VA_ARGS__; } else { constexpr int N = 33; __VA_ARGS__; } } while (0)

// Products of Type-A pairings on the limb-form routines, one TERM per lane (AL::miller_record_lane): the Miller value
// of term t goes to workspace record t; al_prod_finish_kernel then multiplies the k values of each product and runs its
// final exponentiation (one product per lane).
template <int N>
__global__ void __launch_bounds__(kBlock, PBC_A_WAVES) al_miller_kernel(uint4 *ws, const uint8_t *g1, const uint8_t *g2,
                                                                        size_t n, unsigned *ctr, KArgs<N> ka) {
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;
    constexpr int L = 8 * N;
    uint4 rec[AL<N>::MREC];
    AL<N>::miller_record_lane(rec, g1 + ld * L, g2 + ld * L);
    if (idx < n) {
#pragma unroll
      for (int i = 0; i < AL<N>::MREC; i++) ws[idx * AL<N>::MREC + i] = rec[i];
    }
  }
}
template <int N>
__global__ void __launch_bounds__(kBlock, PBC_A_WAVES) al_prod_finish_kernel(uint8_t *gt, const uint4 *ws, size_t n, int k,
                                                                             unsigned *ctr, KArgs<N> ka) {
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;
    constexpr int L = 8 * N;
    __attribute__((aligned(16))) uint8_t out[L];
    AL<N>::prod_finish_lane(out, ws + ld * (size_t) k * AL<N>::MREC, k);
    if (idx < n) {
      uint4 *dst = reinterpret_cast<uint4 *>(gt + idx * L);
      const uint4 *src = reinterpret_cast<const uint4 *>(out);
#pragma unroll
      for (int i = 0; i < L / 16; i++) dst[i] = src[i];
    }
  }
}

// One k-term product of Type-A pairings per lane (terms of unit u are records u*k .. u*k+k-1).  `ws` is the object's
// workspace for the per-term Miller state: k x 24 x 128 uint4 per workgroup (a_prod_pairing_lane).
template <int N>
__global__ void __launch_bounds__(kBlock, PBC_A_WAVES) a_prod_pairing_kernel(uint8_t *gt, const uint8_t *g1,
                                                                 const uint8_t *g2, size_t n, int k, uint4 *ws, unsigned *ctr, KArgs<N> ka) {
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;
    constexpr int L = 8 * N;
    __attribute__((aligned(16))) uint8_t out[L];
    __shared__ uint32_t lds_f[2 * N * kBlock];   // the shared accumulator of every lane, limb-major: conflict-free
    a_prod_pairing_lane<N>(out, g1 + ld * k * L, g2 + ld * k * L, k,
                           ws + (size_t) blockIdx.x * (size_t) k * (6 * (N / 4) * kBlock) + threadIdx.x, lds_f + threadIdx.x, kBlock);
    if (idx < n) {
      uint4 *dst = reinterpret_cast<uint4 *>(gt + idx * L);
      const uint4 *src = reinterpret_cast<const uint4 *>(out);
  #pragma unroll
      for (int i = 0; i < L / 16; i++) dst[i] = src[i];
    }
  }
}

// Type A1: one k-term product (k = 1: a single pairing) per lane; 130-byte coordinates for a1.param.
template <int N>
__global__ void __launch_bounds__(kWide<N>, N >= 32 ? PBC_A1_WAVES : PBC_A_WAVES) a1_prod_pairing_kernel(uint8_t *gt, const uint8_t *g1,
                                                                               const uint8_t *g2, size_t n, int k, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kWide<N> + threadIdx.x;
  size_t ld = idx < n ? idx : n - 1;
  const int L = 2 * fq_bytes<N>();
  __attribute__((aligned(4))) uint8_t out[8 * N];
  // register-resident fields park Q in LDS (limb-major); the 33-word fields keep Q in private memory and the Miller
  // step's two hottest temporaries in LDS, 72 words per lane (pairing_a.cuh a_double_step)
  __shared__ __attribute__((aligned(16))) uint32_t lds_q[kMemOperands<N> ? 72 * kWide<N> : 2 * N * kWide<N>];
  a1_prod_pairing_lane<N>(out, g1 + ld * k * L, g2 + ld * k * L, k, lds_q + threadIdx.x * (kMemOperands<N> ? 72 : 1), kWide<N>);
  if (idx < n) {
    if ((L & 3) == 0) {
      uint32_t *dst = reinterpret_cast<uint32_t *>(gt + idx * L);
      const uint32_t *src = reinterpret_cast<const uint32_t *>(out);
      for (int i = 0; i < L / 4; i++) dst[i] = src[i];
    } else {
      for (int i = 0; i < L; i++) gt[idx * L + i] = out[i];
    }
  }
}

// Type E: one k-term product (k = 1: a single pairing) per lane; G1/G2 256 B, GT 128 B for e.param.
template <int N>
__global__ void __launch_bounds__(kWide<N>, N >= 32 ? PBC_A1_WAVES : PBC_A_WAVES) e_prod_pairing_kernel(uint8_t *gt, const uint8_t *g1,
                                                                              const uint8_t *g2, size_t n, int k, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kWide<N> + threadIdx.x;
  size_t ld = idx < n ? idx : n - 1;
  const int LT = fq_bytes<N>(), L = 2 * LT;
  __attribute__((aligned(4))) uint8_t out[4 * N];
  // 33-word fields: the two hottest elements of a step (pairing_e.cuh ejac) in LDS, 72 words per lane
  __shared__ __attribute__((aligned(16))) uint32_t lds_hot[kMemOperands<N> ? kWideHotWords * kWide<N> : 4];
  uint32_t *lds_q = kMemOperands<N> ? lds_hot + threadIdx.x * kWideHotWords : nullptr;
  e_prod_pairing_lane<N>(out, g1 + ld * k * L, g2 + ld * k * L, k, lds_q, kWide<N>);
  if (idx < n) {
    if ((LT & 3) == 0) {
      uint32_t *dst = reinterpret_cast<uint32_t *>(gt + idx * LT);
      const uint32_t *src = reinterpret_cast<const uint32_t *>(out);
      for (int i = 0; i < LT / 4; i++) dst[i] = src[i];
    } else {
      for (int i = 0; i < LT; i++) gt[idx * LT + i] = out[i];
    }
  }
}

// pairing_pp_init: ONE lane derives the line-coefficient table of a fixed first argument.
template <int N>
__global__ void a_pp_init_kernel(uint32_t *tab, uint32_t *valid, const uint8_t *g1, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  *valid = a_pp_init_lane<N>(tab, g1) ? 1u : 0u;
}
// pairing_pp_apply over a batch of second arguments, one per lane; the table is uniform data.
template <int N>
__global__ void __launch_bounds__(kBlock, PBC_A_WAVES) al_pp_apply_kernel(uint8_t *gt, const uint32_t *__restrict__ tab,
                                                                           const uint32_t *__restrict__ valid,
                                                                           const uint8_t *g2, size_t n, unsigned *ctr, KArgs<N> ka) {
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;
    constexpr int L = 8 * N;
    __attribute__((aligned(16))) uint8_t out[L];
    AL<N>::pp_apply_lane(out, tab, *valid != 0, g2 + ld * L);
    if (idx < n) {
      uint4 *dst = reinterpret_cast<uint4 *>(gt + idx * L);
      const uint4 *src = reinterpret_cast<const uint4 *>(out);
  #pragma unroll
      for (int i = 0; i < L / 16; i++) dst[i] = src[i];
    }
  }
}

// pairing_pp for type a1
template <int N>
__global__ void a1_pp_init_kernel(uint32_t *tab, uint32_t *valid, const uint8_t *g1, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  *valid = a1_pp_init_lane<N>(tab, g1) ? 1u : 0u;
}
template <int N>
__global__ void __launch_bounds__(kWide<N>, N >= 32 ? PBC_A1_WAVES : PBC_A_WAVES) a1_pp_apply_kernel(uint8_t *gt, const uint32_t *__restrict__ tab,
                                                                           const uint32_t *__restrict__ valid,
                                                                           const uint8_t *g2, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kWide<N> + threadIdx.x;
  size_t ld = idx < n ? idx : n - 1;
  const int L = 2 * fq_bytes<N>();
  __attribute__((aligned(4))) uint8_t out[8 * N];
  __shared__ __attribute__((aligned(16))) uint32_t lds_f[kMemOperands<N> ? 72 * kWide<N> : 4];   // 33-word fields: f^2 of a step in LDS
  a1_pp_apply_lane<N>(out, tab, *valid != 0, g2 + ld * L, kMemOperands<N> ? lds_f + threadIdx.x * 72 : nullptr);
  if (idx < n) {
    if ((L & 3) == 0) {
      uint32_t *dst = reinterpret_cast<uint32_t *>(gt + idx * L);
      const uint32_t *src = reinterpret_cast<const uint32_t *>(out);
      for (int i = 0; i < L / 4; i++) dst[i] = src[i];
    } else {
      for (int i = 0; i < L; i++) gt[idx * L + i] = out[i];
    }
  }
}

template <int N> __global__ void e_init_kernel(EConst *out, ERaw raw, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  e_init_lane<N>(out, raw, c_e);
}

int derive_e(pbc_hip_pairing_s *P, hipStream_t s) {
  DevBuf buf;
  HIP_TRY(buf.alloc(sizeof(EConst)));
  EConst *dbuf = buf.as<EConst>();
  if (P->nlimb == 16) hipLaunchKernelGGL(e_init_kernel<16>, dim3(1), dim3(64), 0, s, dbuf, P->eraw, kargs<16>(P));
  else hipLaunchKernelGGL(e_init_kernel<33>, dim3(1), dim3(64), 0, s, dbuf, P->eraw, kargs<33>(P));
  HIP_TRY(hipMemcpyAsync(&P->econst, dbuf, sizeof(EConst), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  if (P->econst.rxs < 1 || P->econst.rxs > 255)      // the steps multiply by x_R with additions (pairing_e.cuh)
    return fail("type e: no auxiliary point with x in 1..255 on this curve (found x = %d)", P->econst.rxs);
  return 0;
}

// The 33-word kernels are budgeted for TWO waves per SIMD (PBC_A1_WAVES) and launched as 256-lane workgroups (kWide):
// the four waves of a workgroup go to the four SIMDs of a CU.  With 128-lane workgroups the dispatcher put both waves
// of every workgroup on the first SIMDs with room, so a launch that would fit at one wave per SIMD ran two to a SIMD on
// half the SIMDs (e.param, 2^16 units: 62 ms against 48; asking for LDS to limit the workgroups per CU did not move it).
int launch_a(pbc_hip_pairing_s *P, void *d_gt, const void *d_g1, const void *d_g2, size_t n, int k, hipStream_t s, ProdWs &W) {
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  if (P->type == 'a' && !P->a_generic && k == 1 && n <= P->a_wave_max) {
    if (n <= P->a_wave4_max)
      hipLaunchKernelGGL((aw_pairing_kernel<16, 4>), dim3((unsigned) n), dim3(256), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, kargs<16>(P));
    else if (n <= P->a_wave2_max)
      hipLaunchKernelGGL((aw_pairing_kernel<16, 2>), dim3((unsigned) n), dim3(128), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, kargs<16>(P));
    else
      hipLaunchKernelGGL((aw_pairing_kernel<16, 1>), dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, kargs<16>(P));
  } else if (P->type == 'a' && !P->a_generic && k == 1) {
    // A batch a little above a whole number of strides of the resident grid would pay a whole stride for its tail
    // (131073 units: 17.2 ms against 10.3 for 131072): a tail of wave-kernel size goes to the wave kernel instead,
    // behind the lane kernel on the same stream (+1.2 to +5.4 ms).
    const unsigned rg = PBC_RGRID(al_pairing_kernel<16>);
    const size_t stride = (size_t) rg * kBlock, tail = n % stride;
    if (n > stride && tail && tail <= P->a_wave_max) {
      const size_t head = n - tail;
      hipLaunchKernelGGL(al_pairing_kernel<16>, dim3(rg), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                         (const uint8_t *) d_g1, (const uint8_t *) d_g2, head, unit_counter(P, s), kargs<16>(P));
      HIP_TRY(hipGetLastError());
      return launch_a(P, (uint8_t *) d_gt + head * P->lenT, (const uint8_t *) d_g1 + head * P->len1, (const uint8_t *) d_g2 + head * P->len2, tail, 1, s, W);
    }
    hipLaunchKernelGGL(al_pairing_kernel<16>, dim3(rg), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                       (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, unit_counter(P, s), kargs<16>(P));
  } else if (P->type == 'a' && !P->a_generic && k > 1 && n * (size_t) k <= P->a_wave_max && !P->a_prod_shared && P->a_prod_chunk == kProdChunkDefault) {
    // ("hip_prod_shared 1" / "hip_prod_chunk N" name a kernel explicitly: those objects never take the wave routines)
    // a few terms in all (benchmark/multipairing.c's shape, or one element_prod_pairing through the hooks): a wave (four,
    // up to hip_wave4_max terms) per TERM, then a wave (four) per PRODUCT -- the latency of one Miller loop and one final
    // exponentiation on the wave routines instead of the 6.4 ms of a lane kernel (pairing_aw.cuh)
    const size_t nt = n * (size_t) k;
    uint32_t *ws = (uint32_t *) W.get(nt * AW<16, 1>::WREC * sizeof(uint32_t));
    if (!ws) return 1;
    if (nt <= P->a_wave4_max) {
      hipLaunchKernelGGL((aw_miller_kernel<16, 4>), dim3((unsigned) nt), dim3(256), 0, s, ws, (const uint8_t *) d_g1, (const uint8_t *) d_g2, kargs<16>(P));
      hipLaunchKernelGGL((aw_prod_finish_kernel<16, 4>), dim3((unsigned) n), dim3(256), 0, s, (uint8_t *) d_gt, (const uint32_t *) ws, k, kargs<16>(P));
    } else {
      hipLaunchKernelGGL((aw_miller_kernel<16, 1>), dim3((unsigned) nt), dim3(64), 0, s, ws, (const uint8_t *) d_g1, (const uint8_t *) d_g2, kargs<16>(P));
      hipLaunchKernelGGL((aw_prod_finish_kernel<16, 1>), dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint32_t *) ws, k, kargs<16>(P));
    }
  } else if (P->type == 'a' && !P->a_generic && !P->a_prod_shared) {
    // one term per lane, then one product per lane; at most a_prod_chunk terms in flight (their records: 160 B each)
    const size_t per = std::max<size_t>(1, P->a_prod_chunk / (size_t) k);
    const size_t first = std::min(n, per);
    void *ws = W.get(first * (size_t) k * AL<16>::MREC * sizeof(uint4));
    if (!ws) return 1;
    for (size_t u0 = 0; u0 < n; u0 += per) {
      const size_t nu = std::min(per, n - u0), nt = nu * (size_t) k;
      hipLaunchKernelGGL(al_miller_kernel<16>, dim3(resident_grid(P, reinterpret_cast<const void *>(&al_miller_kernel<16>), nt)), dim3(kBlock), 0, s,
                         (uint4 *) ws, (const uint8_t *) d_g1 + u0 * (size_t) k * P->len1, (const uint8_t *) d_g2 + u0 * (size_t) k * P->len2, nt, unit_counter(P, s), kargs<16>(P));
      hipLaunchKernelGGL(al_prod_finish_kernel<16>, dim3(resident_grid(P, reinterpret_cast<const void *>(&al_prod_finish_kernel<16>), nu)), dim3(kBlock), 0, s,
                         (uint8_t *) d_gt + u0 * P->lenT, (const uint4 *) ws, nu, k, unit_counter(P, s), kargs<16>(P));
    }
  } else if (P->type == 'a' && !P->a_generic) {
    grid = PBC_RGRID(a_prod_pairing_kernel<16>);                      // one workspace record per RESIDENT workgroup
    void *ws = W.get((size_t) grid * (size_t) k * (6 * 4 * kBlock) * sizeof(uint4));
    if (!ws) return 1;
    hipLaunchKernelGGL(a_prod_pairing_kernel<16>, dim3(grid), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                       (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, k, (uint4 *) ws, unit_counter(P, s), kargs<16>(P));
  } else if (ag_capable(P) && n <= P->ag_wave_max && n * (size_t) k <= ((size_t) 1 << 20)) {
    // small batches of type a1 / generic type a: a wave (four, up to hip_wave4_max terms) per pairing -- per TERM of a product, then
    // one per product -- on the wave routines of the signed-digit Miller loop (a lane's time grows with the terms of its product
    // as the wavefronts' does: the cut-over is a number of products)
    const uint32_t *aux = ag_device_aux(P);
    if (!aux) return 1;
    const size_t nt = n * (size_t) k;
    const bool four = nt <= P->ag_wave4_max, eight = nt <= P->ag_wave8_max;
    if (k == 1) {
      PBC_DISPATCH_AG(P, {
        if (eight) hipLaunchKernelGGL((agw_pairing_kernel<N, 8>), dim3((unsigned) n), dim3(512), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
        else if (four) hipLaunchKernelGGL((agw_pairing_kernel<N, 4>), dim3((unsigned) n), dim3(256), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
        else hipLaunchKernelGGL((agw_pairing_kernel<N, 1>), dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
      });
    } else {
      uint32_t *ws = (uint32_t *) W.get(nt * AW<33, 1, AG<33>>::WREC * sizeof(uint32_t));
      if (!ws) return 1;
      PBC_DISPATCH_AG(P, {
        if (eight) {
          hipLaunchKernelGGL((agw_miller_kernel<N, 8>), dim3((unsigned) nt), dim3(512), 0, s, ws, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
          hipLaunchKernelGGL((agw_prod_finish_kernel<N, 4>), dim3((unsigned) n), dim3(256), 0, s, (uint8_t *) d_gt, (const uint32_t *) ws, k, aux, kargs<N>(P));
        } else if (four) {
          hipLaunchKernelGGL((agw_miller_kernel<N, 4>), dim3((unsigned) nt), dim3(256), 0, s, ws, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
          hipLaunchKernelGGL((agw_prod_finish_kernel<N, 4>), dim3((unsigned) n), dim3(256), 0, s, (uint8_t *) d_gt, (const uint32_t *) ws, k, aux, kargs<N>(P));
        } else {
          hipLaunchKernelGGL((agw_miller_kernel<N, 1>), dim3((unsigned) nt), dim3(64), 0, s, ws, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
          hipLaunchKernelGGL((agw_prod_finish_kernel<N, 1>), dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint32_t *) ws, k, aux, kargs<N>(P));
        }
      });
    }
  } else if ((P->type == 'a' || P->type == '1') && P->nlimb == 16) {   // other sizes: the bit-by-bit kernels
    hipLaunchKernelGGL(a1_prod_pairing_kernel<16>, dim3(grid), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                       (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, k, kargs<16>(P));
  } else if (P->type == '1' || P->type == 'a') {
    hipLaunchKernelGGL(a1_prod_pairing_kernel<33>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, (uint8_t *) d_gt,
                       (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, k, kargs<33>(P));
  } else if (P->type == 'e' && !P->ag_aux.empty() && n <= P->ag_wave_max && n * (size_t) k <= ((size_t) 1 << 20)) {
    // small batches of type e: a workgroup of four wavefronts (one, above hip_wave4_max terms) per pairing / per TERM, then one per product
    const uint32_t *aux = ag_device_aux(P);
    if (!aux) return 1;
    const size_t nt = n * (size_t) k;
    const bool four = nt <= P->ag_wave4_max;
    if (k == 1) {
      PBC_DISPATCH_AG(P, {
        if (four) hipLaunchKernelGGL((ew_pairing_kernel<N, 4>), dim3((unsigned) n), dim3(256), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
        else hipLaunchKernelGGL((ew_pairing_kernel<N, 1>), dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
      });
    } else {
      uint32_t *ws = (uint32_t *) W.get(nt * EW<33, 1>::WREC * sizeof(uint32_t));
      if (!ws) return 1;
      PBC_DISPATCH_AG(P, {
        if (four) {
          hipLaunchKernelGGL((ew_miller_kernel<N, 4>), dim3((unsigned) nt), dim3(256), 0, s, ws, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
          hipLaunchKernelGGL((ew_prod_finish_kernel<N, 4>), dim3((unsigned) n), dim3(256), 0, s, (uint8_t *) d_gt, (const uint32_t *) ws, k, aux, kargs<N>(P));
        } else {
          hipLaunchKernelGGL((ew_miller_kernel<N, 1>), dim3((unsigned) nt), dim3(64), 0, s, ws, (const uint8_t *) d_g1, (const uint8_t *) d_g2, aux, kargs<N>(P));
          hipLaunchKernelGGL((ew_prod_finish_kernel<N, 1>), dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint32_t *) ws, k, aux, kargs<N>(P));
        }
      });
    }
  } else if (P->type == 'e' && P->nlimb == 16) {
    hipLaunchKernelGGL(e_prod_pairing_kernel<16>, dim3(grid), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                       (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, k, kargs<16>(P));
  } else if (P->type == 'e') {
    hipLaunchKernelGGL(e_prod_pairing_kernel<33>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, (uint8_t *) d_gt,
                       (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, k, kargs<33>(P));
  } else {
    return fail("unsupported type");
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

void pp_init_launch_a(pbc_hip_pairing_s *P, pbc_hip_pp_s *pp, const uint8_t *dg1, bool a1) {
  if (a1 && P->nlimb == 16) {
    hipLaunchKernelGGL(a1_pp_init_kernel<16>, dim3(1), dim3(64), 0, 0, pp->tab, pp->valid, dg1, kargs<16>(P));
  } else if (a1) {
    hipLaunchKernelGGL(a1_pp_init_kernel<33>, dim3(1), dim3(64), 0, 0, pp->tab, pp->valid, dg1, kargs<33>(P));
  } else {
    hipLaunchKernelGGL(a_pp_init_kernel<16>, dim3(1), dim3(64), 0, 0, pp->tab, pp->valid, dg1, kargs<16>(P));
  }
}
int pp_apply_launch_a(pbc_hip_pp_s *pp, void *d_gt, const void *d_g2, size_t n, hipStream_t s) {
  pbc_hip_pairing_s *P = pp->P;
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  if (P->type == 'a' && !P->a_generic && n <= P->a_wave_max) {
    // small batches (benchmark/benchmark.c:75-81 times pairing_pp_apply one at a time): a wave -- four, up to
    // hip_wave4_max units -- per second argument (pairing_aw.cuh pp_apply_wave)
    if (n <= P->a_wave4_max)
      hipLaunchKernelGGL((aw_pp_apply_kernel<16, 4>), dim3((unsigned) n), dim3(256), 0, s, (uint8_t *) d_gt, pp->tab, pp->valid, (const uint8_t *) d_g2, kargs<16>(P));
    else if (n <= P->a_wave2_max)
      hipLaunchKernelGGL((aw_pp_apply_kernel<16, 2>), dim3((unsigned) n), dim3(128), 0, s, (uint8_t *) d_gt, pp->tab, pp->valid, (const uint8_t *) d_g2, kargs<16>(P));
    else
      hipLaunchKernelGGL((aw_pp_apply_kernel<16, 1>), dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, pp->tab, pp->valid, (const uint8_t *) d_g2, kargs<16>(P));
  } else if (P->type == 'a' && !P->a_generic) {
    hipLaunchKernelGGL(al_pp_apply_kernel<16>, dim3(PBC_RGRID(al_pp_apply_kernel<16>)), dim3(kBlock), 0, s, (uint8_t *) d_gt, pp->tab, pp->valid,
                       (const uint8_t *) d_g2, n, unit_counter(P, s), kargs<16>(P));
  } else if (ag_capable(P) && n <= P->ag_wave_max) {
    const uint32_t *aux = ag_device_aux(P);
    if (!aux) return 1;
    PBC_DISPATCH_AG(P, {
      if (n <= P->ag_wave4_max) hipLaunchKernelGGL((agw_pp_apply_kernel<N, 4>), dim3((unsigned) n), dim3(256), 0, s, (uint8_t *) d_gt, pp->tab, pp->valid, (const uint8_t *) d_g2, aux, kargs<N>(P));
      else hipLaunchKernelGGL((agw_pp_apply_kernel<N, 1>), dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, pp->tab, pp->valid, (const uint8_t *) d_g2, aux, kargs<N>(P));
    });
  } else if (P->nlimb == 16) {
    hipLaunchKernelGGL(a1_pp_apply_kernel<16>, dim3(grid), dim3(kBlock), 0, s, (uint8_t *) d_gt, pp->tab, pp->valid,
                       (const uint8_t *) d_g2, n, kargs<16>(P));
  } else {
    hipLaunchKernelGGL(a1_pp_apply_kernel<33>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, (uint8_t *) d_gt, pp->tab, pp->valid,
                       (const uint8_t *) d_g2, n, kargs<33>(P));
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
