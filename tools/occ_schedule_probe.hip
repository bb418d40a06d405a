// Per-wave schedule probe (hipcc --offload-arch=gfx950 -O2 tools/occ_schedule_probe.hip -o /tmp/occ2): start / end timestamps and HW_ID of every wave of a kernel that mimics a footprint (LDS, scratch, VGPRs, kernarg size) -- which workgroups start when, mean residency per SIMD (profiles/r03_notes.md)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
struct Big { uint32_t w[760]; };   // 3040-byte kernarg
template <int LDS_BYTES, int SCRATCH_BYTES, int NV, int WAVES>
__global__ void __launch_bounds__(128, WAVES) occ(uint64_t *out, uint64_t *trace, uint32_t a, uint32_t b, int iters, int idx, Big kb) {
  __shared__ uint32_t lds[LDS_BYTES / 4 + 1];
  volatile uint32_t priv[SCRATCH_BYTES / 4 + 1];
  uint64_t t0 = __builtin_readcyclecounter();
  uint64_t t0r = wall_clock64();
  uint32_t hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  priv[idx & (SCRATCH_BYTES / 4)] = a;
  if (SCRATCH_BYTES) priv[(idx * 7) % (SCRATCH_BYTES / 4 + 1)] = b;
  lds[(threadIdx.x + idx) % (LDS_BYTES / 4 + 1)] = b + kb.w[idx & 511];
  __syncthreads();
  uint32_t r[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) r[i] = a * (i + 1) + threadIdx.x;
  uint64_t acc = threadIdx.x;
  uint32_t x = a + threadIdx.x + lds[(threadIdx.x * 3 + idx) % (LDS_BYTES / 4 + 1)], y = b ^ priv[(idx * 3) % (SCRATCH_BYTES / 4 + 1)];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 48; q++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
#pragma unroll
    for (int i = 0; i < NV; i++) asm volatile("" : "+v"(r[i]));
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < NV; i++) s += r[i];
  out[blockIdx.x * 128 + threadIdx.x] = acc + priv[idx % (SCRATCH_BYTES / 4 + 1)] + s;
  uint64_t t1r = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    uint64_t *t = trace + (size_t) (blockIdx.x * 2 + (threadIdx.x >> 6)) * 4;
    t[0] = t0r; t[1] = t1r; t[2] = hw; t[3] = xcc;
  }
}
template <class K>
static void run(const char *name, K kern, int blocks = 2048) {
  const int iters = 4000;
  uint64_t *d, *tr;
  hipMalloc(&d, (size_t) blocks * 128 * 8);
  hipMalloc(&tr, (size_t) blocks * 2 * 4 * 8);
  Big kb = {};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(128), 0, 0, d, tr, 12345u, 67890u, iters, rep, kb);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<uint64_t> h((size_t) blocks * 2 * 4);
  hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
  uint64_t tmin = ~0ull, tmax = 0;
  double sum = 0;
  for (int w = 0; w < blocks * 2; w++) { tmin = std::min(tmin, h[w * 4]); tmax = std::max(tmax, h[w * 4 + 1]); sum += (double) (h[w * 4 + 1] - h[w * 4]); }
  // residency: sum of wave durations / (kernel span * 1024 SIMDs)
  const double span = (double) (tmax - tmin);
  // how many waves start within the first 2 % of the span (the first "round")
  int first = 0;
  for (int w = 0; w < blocks * 2; w++) if (h[w * 4] - tmin < 0.02 * span) first++;
  // distinct (xcc, se, sh?, cu, simd) keys
  std::vector<uint32_t> keys;
  for (int w = 0; w < blocks * 2; w++) keys.push_back((uint32_t) (h[w * 4 + 3] & 0xf) << 16 | (uint32_t) (h[w * 4 + 2] & 0xfff0));
  std::sort(keys.begin(), keys.end());
  const int distinct = (int) (std::unique(keys.begin(), keys.end()) - keys.begin());
  int hs[10] = {0}, he[10] = {0};
  for (int w = 0; w < blocks * 2; w++) { hs[std::min(9, (int) (10.0 * (h[w * 4] - tmin) / span))]++; he[std::min(9, (int) (10.0 * (h[w * 4 + 1] - tmin) / span))]++; }
  printf("   starts:"); for (int i = 0; i < 10; i++) printf(" %4d", hs[i]); printf("\n   ends:  "); for (int i = 0; i < 10; i++) printf(" %4d", he[i]); printf("\n");
  printf("%-28s %7.2f ms  span %.0f ticks  mean residency %.2f waves/SIMD  waves started in the first 2%%: %d  distinct SIMD keys %d\n", name, ms, span,
         sum / span / 1024.0, first, distinct);
  hipFree(d); hipFree(tr);
}
#define RUN(L, S, NV, W) run(#L "/" #S "/" #NV "/" #W, occ<L, S, NV, W>)
#define RUNB(L, S, NV, W, B) run(#L "/" #S "/" #NV "/" #W " x" #B, occ<L, S, NV, W>, B)
int main() {
  RUN(36864, 0, 230, 2);
  RUN(0, 0, 230, 2);
  RUN(39936, 0, 230, 2);
  RUN(32768, 0, 230, 2);
  RUN(20480, 0, 230, 2);
  RUN(8192, 0, 230, 2);
  RUNB(36864, 0, 230, 2, 1024);
  RUNB(36864, 0, 230, 2, 4096);
  RUNB(36864, 0, 230, 2, 8192);
  RUNB(0, 0, 230, 2, 8192);
  return 0;
}
