// pbc_hip_d.hip -- kernels and launches of types d and g (libpbc_hip.so; see host_common.h)
#include "host_common.h"
#include "pairing_dw.cuh"
#include "dw_sched.h"
#include "pairing_gw.cuh"
#include "gw_sched.h"

// Types D and G: one k-term product (k = 1: a single pairing) per lane.  With fb = fixed byte length
// of F_q and d = k/2, G1 records are 2 fb, G2 and GT 2 d fb bytes (40 / 120 / 120 B for d159.param,
// 38 / 190 / 190 B for g149.param).
template <int N, int DEG>
static __device__ __forceinline__ void d_prod_unit(size_t idx, uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n, int k, uint32_t *ws) {
  size_t ld = idx < n ? idx : n - 1;
  const int fb = (int) fpk<N>().fbytes, L1 = 2 * fb, L2 = 2 * DEG * fb, LT = 2 * DEG * fb;
  __attribute__((aligned(4))) uint8_t out[8 * DEG * N];
  TypeMNT<N, DEG>::d_prod_pairing_lane(out, g1 + ld * k * L1, g2 + ld * k * L2, k,
                                       ws + (size_t) blockIdx.x * (size_t) k * (TypeMNT<N, DEG>::DL_WORDS * kBlock) + threadIdx.x);
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
// (resident workgroups where they pay: kDResident, pairing_d.cuh)
template <int N, int DEG>
__global__ void __launch_bounds__(kBlock, PBC_DF_WAVES) d_prod_pairing_kernel(uint8_t *gt, const uint8_t *g1,
                                                                 const uint8_t *g2, size_t n, int k, uint32_t *ws, unsigned *ctr, KArgs<N> ka) {
  if constexpr (kDResident<N, DEG>) {
    PBC_RESIDENT_LOOP(n, ctr) d_prod_unit<N, DEG>(PBC_UNIT_INDEX, gt, g1, g2, n, k, ws);
  } else {
    d_prod_unit<N, DEG>((size_t) blockIdx.x * kBlock + threadIdx.x, gt, g1, g2, n, k, ws);
  }
}

// small batches on the five-word d = 3 fields: one pairing per wavefront (pairing_dw.cuh)
static constexpr size_t kDwMaxTerms = (size_t) 1 << 20;       // (records of the products' workspace: 160 bytes each)
template <int N>
__global__ void __launch_bounds__(64) dw_pairing_kernel(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= n) return;
  const size_t fb = fpk<N>().fbytes;
  DW<N>::pairing(gt + idx * 6 * fb, g1 + idx * 2 * fb, g2 + idx * 6 * fb, sched);
}
// element_prod_pairing on wavefronts: the Miller value of every TERM (n k wavefronts), then one wavefront per product
template <int N>
__global__ void __launch_bounds__(64) dw_miller_kernel(uint32_t *recs, const uint8_t *g1, const uint8_t *g2, size_t terms, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= terms) return;
  const size_t fb = fpk<N>().fbytes;
  DW<N>::miller_term(recs + idx * DW<N>::kRec, g1 + idx * 2 * fb, g2 + idx * 6 * fb, sched);
}
template <int N>
__global__ void __launch_bounds__(64) dw_finish_kernel(uint8_t *gt, const uint32_t *recs, size_t n, int k, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= n) return;
  DW<N>::finish(gt + idx * 6 * fpk<N>().fbytes, recs + idx * (size_t) k * DW<N>::kRec, k, sched);
}
// pairing_pp_apply on wavefronts
template <int N>
__global__ void __launch_bounds__(64) dw_pp_apply_kernel(uint8_t *gt, const uint32_t *__restrict__ tab, const uint32_t *__restrict__ valid, const uint8_t *g2,
                                                         size_t n, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= n) return;
  const size_t fb = fpk<N>().fbytes;
  DW<N>::pp_apply(gt + idx * 6 * fb, tab, *valid != 0, g2 + idx * 6 * fb, sched);
}
// the schedules for this object's curve (dw_sched.h), built on first use and kept with the object
static const DwSched &dw_schedules(pbc_hip_pairing_s *P) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (P->dw_sched.e.empty()) {
    const DConst &C = P->dconst;
    dw::build_schedules(P->dw_sched, C.rbits, [&C](int m) { return (int) ((C.r[m >> 5] >> (m & 31)) & 1) - (int) ((C.rm[m >> 5] >> (m & 31)) & 1); },
                        C.phik, C.phikbits);
  }
  return P->dw_sched;
}
// (five-, six- and seven-word fields: six, seven and eight limbs -- the level programs do not know the field)
static bool dw_capable(const pbc_hip_pairing_s *P) { return P->type == 'd' && P->nlimb >= 5 && P->nlimb <= 7 && P->deg == 3; }
#define PBC_DISPATCH_DW(P, ...) do { if ((P)->nlimb == 5) { constexpr int N = 5; __VA_ARGS__; } else if ((P)->nlimb == 6) { constexpr int N = 6; __VA_ARGS__; } \
                                     else { constexpr int N = 7; __VA_ARGS__; } } while (0)
// (24 KB, read-only, one copy per device the object runs on -- uploaded on first use, kept with the object)
static const uint64_t *dw_device_schedules(pbc_hip_pairing_s *P, const DwSched **host) {
  const DwSched &S = dw_schedules(P);
  static const char kSchedKey = 0;
  bool fresh = false;
  uint64_t *d_sched = (uint64_t *) object_scratch(P, &kSchedKey, S.e.size() * sizeof(uint64_t), &fresh);
  if (!d_sched) return nullptr;
  if (fresh && hipMemcpy(d_sched, S.e.data(), S.e.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  *host = &S;
  return d_sched;
}
extern "C" size_t pbc_hip_diag_dw_schedule(pbc_hip_pairing_t *P, int which, uint64_t *out, size_t cap) {
  if (!P || !dw_capable(P) || which < 0 || which > 3) return 0;
  const DwSched &S = dw_schedules(P);
  const size_t first = S.off[which], last = which < 3 ? S.off[which + 1] : S.e.size();
  for (size_t i = first; i < last && i - first < cap; i++) out[i - first] = S.e[i];
  return last - first;
}

// small batches of type g on the five-word field: one pairing per wavefront (pairing_gw.cuh)
template <int N>
__global__ void __launch_bounds__(64) gw_pairing_kernel(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= n) return;
  const size_t fb = fpk<N>().fbytes;
  GW<N>::pairing(gt + idx * 10 * fb, g1 + idx * 2 * fb, g2 + idx * 10 * fb, sched);
}
// element_prod_pairing: the Miller value of every TERM (n k wavefronts), then one wavefront per product
template <int N>
__global__ void __launch_bounds__(64) gw_miller_kernel(uint32_t *recs, const uint8_t *g1, const uint8_t *g2, size_t terms, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= terms) return;
  const size_t fb = fpk<N>().fbytes;
  GW<N>::miller_term(recs + idx * GW<N>::kRec, g1 + idx * 2 * fb, g2 + idx * 10 * fb, sched);
}
template <int N>
__global__ void __launch_bounds__(64) gw_finish_kernel(uint8_t *gt, const uint32_t *recs, size_t n, int k, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= n) return;
  GW<N>::finish(gt + idx * 10 * fpk<N>().fbytes, recs + idx * (size_t) k * GW<N>::kRec, k, sched);
}
// pairing_pp_apply
template <int N>
__global__ void __launch_bounds__(64) gw_pp_apply_kernel(uint8_t *gt, const uint32_t *__restrict__ tab, const uint32_t *__restrict__ valid, const uint8_t *g2,
                                                         size_t n, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= n) return;
  const size_t fb = fpk<N>().fbytes;
  GW<N>::pp_apply(gt + idx * 10 * fb, tab, *valid != 0, g2 + idx * 10 * fb, sched);
}
static bool gw_capable(const pbc_hip_pairing_s *P) { return P->type == 'g' && P->nlimb == 5 && P->deg == 5; }
static const DwSched &gw_schedules(pbc_hip_pairing_s *P) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (P->gw_sched.e.empty()) {
    const DConst &C = P->dconst;
    if (!gw::build_schedules(P->gw_sched, C.rbits, [&C](int m) { return (int) ((C.r[m >> 5] >> (m & 31)) & 1) - (int) ((C.rm[m >> 5] >> (m & 31)) & 1); }, C.phik, C.phikbits))
      P->gw_sched.e.clear();
  }
  return P->gw_sched;
}
// (the four schedules: ~50 KB, read-only, one copy per device the object runs on)
static const uint64_t *gw_device_schedules(pbc_hip_pairing_s *P, const DwSched **host) {
  const DwSched &S = gw_schedules(P);
  if (S.e.empty()) return nullptr;
  static const char kGwSchedKey = 0;
  bool fresh = false;
  uint64_t *d_sched = (uint64_t *) object_scratch(P, &kGwSchedKey, S.e.size() * sizeof(uint64_t), &fresh);
  if (!d_sched) return nullptr;
  if (fresh && hipMemcpy(d_sched, S.e.data(), S.e.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  *host = &S;
  return d_sched;
}
extern "C" size_t pbc_hip_diag_gw_schedule(pbc_hip_pairing_t *P, int which, uint64_t *out, size_t cap) {
  if (!P || !gw_capable(P) || which < 0 || which > 3) return 0;
  const DwSched &S = gw_schedules(P);
  if (S.e.empty()) return 0;
  const size_t first = S.off[which], last = which < 3 ? S.off[which + 1] : S.e.size();
  for (size_t i = first; i < last && i - first < cap; i++) out[i - first] = S.e[i];
  return last - first;
}

// pairing_pp for types d / g: single-lane table derivation, then one second argument per lane
template <int N, int DEG>
__global__ void d_pp_init_kernel(uint32_t *tab, uint32_t *valid, const uint8_t *g1, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  *valid = TypeMNT<N, DEG>::d_pp_init_lane(tab, g1) ? 1u : 0u;
}
template <int N, int DEG>
static __device__ __forceinline__ void d_pp_unit(size_t idx, uint8_t *gt, const uint32_t *__restrict__ tab, const uint32_t *__restrict__ valid,
                                                 const uint8_t *g2, size_t n) {
  size_t ld = idx < n ? idx : n - 1;
  const int fb = (int) fpk<N>().fbytes, L2 = 2 * DEG * fb, LT = 2 * DEG * fb;
  __attribute__((aligned(4))) uint8_t out[8 * DEG * N];
  TypeMNT<N, DEG>::d_pp_apply_lane(out, tab, *valid != 0, g2 + ld * L2);
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
template <int N, int DEG>
__global__ void __launch_bounds__(kBlock, PBC_DF_WAVES) d_pp_apply_kernel(uint8_t *gt, const uint32_t *__restrict__ tab,
                                                                          const uint32_t *__restrict__ valid,
                                                                          const uint8_t *g2, size_t n, unsigned *ctr, KArgs<N> ka) {
  if constexpr (kDResident<N, DEG>) {
    PBC_RESIDENT_LOOP(n, ctr) d_pp_unit<N, DEG>(PBC_UNIT_INDEX, gt, tab, valid, g2, n);
  } else {
    d_pp_unit<N, DEG>((size_t) blockIdx.x * kBlock + threadIdx.x, gt, tab, valid, g2, n);
  }
}

template <int N, int DEG> __global__ void d_init_stage1(DConst *out, DRaw raw, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  TypeMNT<N, DEG>::init_stage1(out, raw, c_d);
}
template <int N, int DEG> __global__ void d_init_stage2(DConst *out, DRaw raw, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  TypeMNT<N, DEG>::init_stage2(out, raw);
}

int derive_d(pbc_hip_pairing_s *P, hipStream_t s) {
  DevBuf buf;
  HIP_TRY(buf.alloc(sizeof(DConst)));
  DConst *dbuf = buf.as<DConst>();
  PBC_DISPATCH_D(P, hipLaunchKernelGGL((d_init_stage1<N, DEG>), dim3(1), dim3(64), 0, s, dbuf, P->draw, kargs<N>(P)));
  HIP_TRY(hipMemcpyAsync(&P->dconst, dbuf, sizeof(DConst), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  PBC_DISPATCH_D(P, hipLaunchKernelGGL((d_init_stage2<N, DEG>), dim3(1), dim3(64), 0, s, dbuf, P->draw, kargs<N>(P)));
  HIP_TRY(hipMemcpyAsync(&P->dconst, dbuf, sizeof(DConst), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_d(pbc_hip_pairing_s *P, void *d_gt, const void *d_g1, const void *d_g2, size_t n, int k, hipStream_t s, ProdWs &W) {
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  // The throughput kernel runs a batch this small at the latency of ONE lane (3.9 ms a pairing, ~2.9 ms more per further term
  // of a product): a wavefront per pairing -- per TERM for products, then one per product -- instead (pairing_dw.cuh).
  // type g: 14.5 ms a pairing on one lane whatever the batch size (+ 6.5 ms per further term of a product); a wavefront per pairing
  // -- per TERM for products, then one per product -- instead (the schedules: kept with the object).  Saturated, the wavefronts
  // finish 430 k pairings or 540 k terms a second: products of many terms reach the lane kernel's time a little earlier
  // (measured: 4 terms ~4600 products, 16 terms ~3800).
  const size_t gw_max = k <= 2 ? P->d_wave_max : (size_t) ((double) P->d_wave_max * (0.7 + 0.8 / k));
  if (gw_capable(P) && k >= 1 && n <= gw_max && n * (size_t) k <= kDwMaxTerms) {
    const DwSched *S = nullptr;
    const uint64_t *d_sched = gw_device_schedules(P, &S);
    if (!d_sched) return 1;
    if (k == 1) {
      hipLaunchKernelGGL(gw_pairing_kernel<5>, dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, d_sched + S->off[gw::SCHED_PAIRING], kargs<5>(P));
    } else {
      const size_t terms = n * (size_t) k;
      uint32_t *recs = (uint32_t *) W.get(terms * GW<5>::kRec * sizeof(uint32_t));
      if (!recs) return 1;
      hipLaunchKernelGGL(gw_miller_kernel<5>, dim3((unsigned) terms), dim3(64), 0, s, recs, (const uint8_t *) d_g1, (const uint8_t *) d_g2, terms, d_sched + S->off[gw::SCHED_MILLER], kargs<5>(P));
      hipLaunchKernelGGL(gw_finish_kernel<5>, dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint32_t *) recs, n, k, d_sched + S->off[gw::SCHED_FINISH], kargs<5>(P));
    }
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const bool waves = dw_capable(P) && k >= 1 && n <= P->d_wave_max && n * (size_t) k <= kDwMaxTerms;
  if (waves && k == 1) {
    const DwSched *S = nullptr;
    const uint64_t *d_sched = dw_device_schedules(P, &S);
    if (!d_sched) return 1;
    PBC_DISPATCH_DW(P, hipLaunchKernelGGL(dw_pairing_kernel<N>, dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, d_sched + S->off[dw::SCHED_PAIRING], kargs<N>(P)));
  } else if (waves) {
    const DwSched *S = nullptr;
    const uint64_t *d_sched = dw_device_schedules(P, &S);
    if (!d_sched) return 1;
    const size_t terms = n * (size_t) k;
    uint32_t *recs = (uint32_t *) W.get(terms * DW<7>::kRec * sizeof(uint32_t));
    if (!recs) return 1;
    PBC_DISPATCH_DW(P, {
      hipLaunchKernelGGL(dw_miller_kernel<N>, dim3((unsigned) terms), dim3(64), 0, s, recs, (const uint8_t *) d_g1, (const uint8_t *) d_g2, terms, d_sched + S->off[dw::SCHED_MILLER], kargs<N>(P));
      hipLaunchKernelGGL(dw_finish_kernel<N>, dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint32_t *) recs, n, k, d_sched + S->off[dw::SCHED_FINISH], kargs<N>(P));
    });
  } else if (k == 1) {
    PBC_DISPATCH_D(P, hipLaunchKernelGGL((d_prod_pairing_kernel<N, DEG>), dim3(kDResident<N, DEG> ? PBC_RGRID(d_prod_pairing_kernel<N, DEG>) : grid), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                                                (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, 1, (uint32_t *) nullptr, kDResident<N, DEG> ? unit_counter(P, s) : nullptr, kargs<N>(P)));
  } else {
    size_t rec = 0;                    // words of Miller state per term and lane (the kernel's own constant)
    // (5-word field: products keep one workgroup per 128 units -- the resident shape measured 160 -> 185 ms on 16-term
    // products -- and run without the time-sliced priorities that go with it)
    bool res = false;
    PBC_DISPATCH_D(P, { rec = (size_t) TypeMNT<N, DEG>::DL_WORDS; res = kDResident<N, DEG> && N != 5; if (res) grid = PBC_RGRID(d_prod_pairing_kernel<N, DEG>); });      // one workspace record per RESIDENT workgroup
    void *ws = W.get((size_t) grid * (size_t) k * rec * kBlock * sizeof(uint32_t));
    if (!ws) return 1;
    PBC_DISPATCH_D(P, hipLaunchKernelGGL((d_prod_pairing_kernel<N, DEG>), dim3(grid), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                                                (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, k, (uint32_t *) ws, res ? unit_counter(P, s) : nullptr, kargs<N>(P, false, !res)));
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int pp_init_launch_d(pbc_hip_pairing_s *P, pbc_hip_pp_s *pp, const uint8_t *dg1) {
  PBC_DISPATCH_D(P, hipLaunchKernelGGL((d_pp_init_kernel<N, DEG>), dim3(1), dim3(64), 0, 0, pp->tab, pp->valid, dg1, kargs<N>(P)));
  return 0;
}
int pp_apply_launch_d(pbc_hip_pp_s *pp, void *d_gt, const void *d_g2, size_t n, hipStream_t s) {
  pbc_hip_pairing_s *P = pp->P;
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  if (gw_capable(P) && n <= P->d_wave_max) {
    const DwSched *S = nullptr;
    const uint64_t *d_sched = gw_device_schedules(P, &S);
    if (!d_sched || S->lines > gw::kMaxLines) return 1;
    hipLaunchKernelGGL(gw_pp_apply_kernel<5>, dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint32_t *) pp->tab, (const uint32_t *) pp->valid, (const uint8_t *) d_g2, n,
                       d_sched + S->off[gw::SCHED_PP], kargs<5>(P));
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (dw_capable(P) && n <= P->d_wave_max) {
    const DwSched *S = nullptr;
    const uint64_t *d_sched = dw_device_schedules(P, &S);
    if (!d_sched || S->lines > dw::kMaxLines) return 1;
    PBC_DISPATCH_DW(P, hipLaunchKernelGGL(dw_pp_apply_kernel<N>, dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint32_t *) pp->tab, (const uint32_t *) pp->valid, (const uint8_t *) d_g2, n,
                                          d_sched + S->off[dw::SCHED_PP], kargs<N>(P)));
    HIP_TRY(hipGetLastError());
    return 0;
  }
  PBC_DISPATCH_D(P, hipLaunchKernelGGL((d_pp_apply_kernel<N, DEG>), dim3(kDResident<N, DEG> ? PBC_RGRID(d_pp_apply_kernel<N, DEG>) : grid), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                                           pp->tab, pp->valid, (const uint8_t *) d_g2, n, kDResident<N, DEG> ? unit_counter(P, s) : nullptr, kargs<N>(P)));
  HIP_TRY(hipGetLastError());
  return 0;
}
