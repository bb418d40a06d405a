// pbc_hip_f.hip -- kernels and launches of type f (libpbc_hip.so; see host_common.h)
#include "host_common.h"
#include "pairing_fw.cuh"
#include "fw_sched.h"

// Type F: one k-term product (k = 1: a single pairing) per lane.  With fb = fixed byte length of F_q:
// G1 2 fb, G2 4 fb, GT 12 fb bytes (40 / 80 / 240 B for f.param).
// (the 5-word field keeps its Miller accumulator in one 36 KB LDS area per workgroup: four workgroups per CU, two waves per
// SIMD; with PBC_F_AREAS=2 in two areas, one wave per SIMD and the register budget that goes with it)
template <int N, bool BM1, bool XS = false>
__global__ void __launch_bounds__(kBlock, N <= 5 ? (PBC_F_AREAS == 1 ? 2 : 1) : PBC_F_WAVES) f_prod_pairing_kernel(uint8_t *gt, const uint8_t *g1,
                                                                 const uint8_t *g2, size_t n, int k, unsigned *ctr, KArgs<N> ka) {
#ifdef PBC_F_WHATIF_TRACE                        // what-if builds only (tools/whatif_time.py --trace): per-wave start / end / HW_ID behind the results
  const uint64_t trace_t0 = wall_clock64();
#endif
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;
    const int fb = (int) fpk<N>().fbytes, L1 = 2 * fb, L2 = 4 * fb, LT = 12 * fb;
    __attribute__((aligned(4))) uint8_t out[48 * N];
    TypeF<N, BM1, XS>::f_prod_pairing_lane(out, g1 + ld * (k < 0 ? 1 : k) * L1, g2 + ld * (k < 0 ? 1 : k) * L2, k < 0 ? 1 : k, k < 0);
    if (idx < n) {
      uint32_t *dst = reinterpret_cast<uint32_t *>(gt + idx * LT);      // LT = 12 fb is a multiple of 4
      const uint32_t *src = reinterpret_cast<const uint32_t *>(out);
      for (int i = 0; i < LT / 4; i++) dst[i] = src[i];
    }
  }
#ifdef PBC_F_WHATIF_TRACE
  if ((threadIdx.x & 63) == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint64_t *t = reinterpret_cast<uint64_t *>(gt + ((n * (size_t) (12 * fpk<N>().fbytes) + 255) & ~(size_t) 255)) + (size_t) (blockIdx.x * 2 + (threadIdx.x >> 6)) * 4;
    t[0] = trace_t0; t[1] = wall_clock64(); t[2] = hw; t[3] = xcc;
  }
#endif
}

template <int N>
__global__ void __launch_bounds__(kBlock) f_debug_kernel(int op, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const int LT = 12 * (int) fpk<N>().fbytes;
  __attribute__((aligned(4))) uint8_t o[48 * N];
  TypeF<N>::f_debug_lane(op, o, a + idx * LT, b + idx * LT);
  for (int i = 0; i < LT; i++) out[idx * LT + i] = o[i];
}

template <int N> __global__ void f_init_stage1(FConst *out, FRaw raw, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  TypeF<N>::init_stage1(out, raw, c_f);
}
template <int N> __global__ void f_init_stage2(FConst *out, FRaw raw, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  TypeF<N>::init_stage2(out, raw);
}
template <int N> __global__ void f_init_stage3(FConst *out, FRaw raw, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  TypeF<N>::init_stage3(out, raw);
}
template <int N> __global__ void f_init_stage4(FConst *out, FRaw raw, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  TypeF<N>::init_stage4(out, raw);
}

int derive_f(pbc_hip_pairing_s *P, hipStream_t s) {
  DevBuf buf;
  HIP_TRY(buf.alloc(sizeof(FConst)));
  FConst *dbuf = buf.as<FConst>();
  PBC_DISPATCH_F(P->nlimb, hipLaunchKernelGGL(f_init_stage1<N>, dim3(1), dim3(64), 0, s, dbuf, P->fraw, kargs<N>(P)));
  HIP_TRY(hipMemcpyAsync(&P->fconst, dbuf, sizeof(FConst), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  PBC_DISPATCH_F(P->nlimb, hipLaunchKernelGGL(f_init_stage2<N>, dim3(1), dim3(64), 0, s, dbuf, P->fraw, kargs<N>(P)));
  HIP_TRY(hipMemcpyAsync(&P->fconst, dbuf, sizeof(FConst), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (P->fraw.e4bits > 0) {          // q = 3 mod 4: the i-basis copy for the pairing kernels
    PBC_DISPATCH_F(P->nlimb, hipLaunchKernelGGL(f_init_stage3<N>, dim3(1), dim3(64), 0, s, dbuf, P->fraw, kargs<N>(P)));
    HIP_TRY(hipMemcpyAsync(&P->fconst_i, dbuf, sizeof(FConst), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    P->f_bm1 = P->fconst_i.bm1 != 0;
    if (P->f_bm1 && P->fraw.xs_try) {   // a sparse xi for the pairing kernels: the i-basis block rewritten for the basis X' = X / c
      PBC_DISPATCH_F(P->nlimb, hipLaunchKernelGGL(f_init_stage4<N>, dim3(1), dim3(64), 0, s, dbuf, P->fraw, kargs<N>(P, true)));
      HIP_TRY(hipMemcpyAsync(&P->fconst_i, dbuf, sizeof(FConst), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// small batches on the five-word BN fields: one pairing per wavefront (pairing_fw.cuh; the parameter file's own basis)
template <int N>
__global__ void __launch_bounds__(64) fw_pairing_kernel(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= n) return;
  const size_t fb = fpk<N>().fbytes;
  FW<N>::pairing(gt + idx * 12 * fb, g1 + idx * 2 * fb, g2 + idx * 4 * fb, sched);
}
// element_prod_pairing on wavefronts: the Miller value of every TERM (n k wavefronts), then one wavefront per product
template <int N>
__global__ void __launch_bounds__(64) fw_miller_kernel(uint32_t *recs, const uint8_t *g1, const uint8_t *g2, size_t terms, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= terms) return;
  const size_t fb = fpk<N>().fbytes;
  FW<N>::miller_term(recs + idx * FW<N>::kRec, g1 + idx * 2 * fb, g2 + idx * 4 * fb, sched);
}
template <int N>
__global__ void __launch_bounds__(64) fw_finish_kernel(uint8_t *gt, const uint32_t *recs, size_t n, int k, const uint64_t *sched, KArgs<N> ka) {
  const size_t idx = blockIdx.x;
  if (idx >= n) return;
  FW<N>::finish(gt + idx * 12 * fpk<N>().fbytes, recs + idx * (size_t) k * FW<N>::kRec, k, sched);
}
static constexpr size_t kFwMaxTerms = (size_t) 1 << 19;       // (records of the products' workspace: 320 bytes each)
static bool fw_capable(const pbc_hip_pairing_s *P) { return P->type == 'f' && P->nlimb == 5 && P->fconst.bn_ok; }
// the schedule for this object's curve (fw_sched.h), built on first use and kept with the object
static const DwSched &fw_schedules(pbc_hip_pairing_s *P) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (P->fw_sched.e.empty()) {
    const FConst &C = P->fconst;
    const uint64_t x = (uint64_t) C.bn_x[0] | (uint64_t) C.bn_x[1] << 32;
    if (!fw::build_schedules(P->fw_sched, C.rbits, [&C](int m) { return (int) ((C.r[m >> 5] >> (m & 31)) & 1) - (int) ((C.rm[m >> 5] >> (m & 31)) & 1); }, x, C.bn_xneg != 0))
      P->fw_sched.e.clear();
  }
  return P->fw_sched;
}
// (39 KB, read-only, one copy per device the object runs on -- uploaded on first use, kept with the object)
static const uint64_t *fw_device_schedules(pbc_hip_pairing_s *P, const DwSched **host) {
  const DwSched &S = fw_schedules(P);
  if (S.e.empty()) return nullptr;
  static const char kSchedKey = 0;
  bool fresh = false;
  uint64_t *d_sched = (uint64_t *) object_scratch(P, &kSchedKey, S.e.size() * sizeof(uint64_t), &fresh);
  if (!d_sched) return nullptr;
  if (fresh && hipMemcpy(d_sched, S.e.data(), S.e.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  *host = &S;
  return d_sched;
}
extern "C" size_t pbc_hip_diag_fw_schedule(pbc_hip_pairing_t *P, int which, uint64_t *out, size_t cap) {
  if (!P || !fw_capable(P) || which < 0 || which > 2) return 0;
  const DwSched &S = fw_schedules(P);
  if (S.e.empty()) return 0;
  const size_t first = S.off[which], last = S.off[which + 1];
  for (size_t i = first; i < last && i - first < cap; i++) out[i - first] = S.e[i];
  return last - first;
}

int launch_f(pbc_hip_pairing_s *P, void *d_gt, const void *d_g1, const void *d_g2, size_t n, int k, hipStream_t s, ProdWs &W) {
  if (k >= 1 && fw_capable(P) && n <= P->f_wave_max && n * (size_t) k <= kFwMaxTerms) {
    // The throughput kernel runs a batch this small at the latency of ONE lane (7 ms a pairing, 5 ms more per further term of a
    // product): a wavefront per pairing -- per TERM for products, then one per product -- instead (pairing_fw.cuh)
    const DwSched *S = nullptr;
    const uint64_t *d_sched = fw_device_schedules(P, &S);
    if (!d_sched) return 1;
    if (k == 1) {
      hipLaunchKernelGGL(fw_pairing_kernel<5>, dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, d_sched + S->off[fw::SCHED_PAIRING], kargs<5>(P));
    } else {
      const size_t terms = n * (size_t) k;
      uint32_t *recs = (uint32_t *) W.get(terms * FW<5>::kRec * sizeof(uint32_t));
      if (!recs) return 1;
      hipLaunchKernelGGL(fw_miller_kernel<5>, dim3((unsigned) terms), dim3(64), 0, s, recs, (const uint8_t *) d_g1, (const uint8_t *) d_g2, terms, d_sched + S->off[fw::SCHED_MILLER], kargs<5>(P));
      hipLaunchKernelGGL(fw_finish_kernel<5>, dim3((unsigned) n), dim3(64), 0, s, (uint8_t *) d_gt, (const uint32_t *) recs, n, k, d_sched + S->off[fw::SCHED_FINISH], kargs<5>(P));
    }
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (P->f_bm1 && P->fconst_i.xs_ok && P->nlimb == 5) {     // ... with a sparse xi
    hipLaunchKernelGGL((f_prod_pairing_kernel<5, true, true>), dim3(PBC_RGRID(f_prod_pairing_kernel<5, true, true>)), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                       (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, k, unit_counter(P, s), kargs<5>(P, true));
  } else if (P->f_bm1) {             // i-basis constants and the instantiation that goes with them
    PBC_DISPATCH_F(P->nlimb, hipLaunchKernelGGL((f_prod_pairing_kernel<N, true>), dim3(PBC_RGRID(f_prod_pairing_kernel<N, true>)), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                       (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, k, unit_counter(P, s), kargs<N>(P, true)));
  } else {
    PBC_DISPATCH_F(P->nlimb, hipLaunchKernelGGL((f_prod_pairing_kernel<N, false>), dim3(PBC_RGRID(f_prod_pairing_kernel<N, false>)), dim3(kBlock), 0, s, (uint8_t *) d_gt,
                       (const uint8_t *) d_g1, (const uint8_t *) d_g2, n, k, unit_counter(P, s), kargs<N>(P)));
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// diagnostics: Miller values before the final exponentiation; single F_q^12 operations
int diag_f_miller(pbc_hip_pairing_s *P, void *dt, const void *d1, const void *d2, size_t n) {
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  PBC_DISPATCH_F(P->nlimb, hipLaunchKernelGGL((f_prod_pairing_kernel<N, false>), dim3(grid), dim3(kBlock), 0, 0, (uint8_t *) dt,
                     (const uint8_t *) d1, (const uint8_t *) d2, n, -1, (unsigned *) nullptr, kargs<N>(P)));
  return 0;
}
int diag_f_op(pbc_hip_pairing_s *P, int stage, void *dt, const void *d1, const void *d2, size_t n) {
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  PBC_DISPATCH_F(P->nlimb, hipLaunchKernelGGL(f_debug_kernel<N>, dim3(grid), dim3(kBlock), 0, 0, stage, (uint8_t *) dt, (const uint8_t *) d1,
                     (const uint8_t *) d2, n, kargs<N>(P)));
  return 0;
}
