// Standalone probe (hipcc --offload-arch=gfx950 -O2 tools/occupancy_probe.hip -o /tmp/occ_probe): how many waves per SIMD
// are really resident for a 128-lane workgroup with a given LDS footprint and private-memory (scratch) size per lane?
// One wave alone gets a v_mad_u64_u32 through every ~9.1 cycles, two or more share the pipe at ~4.6 cycles
// (tools/mac_chain_probe.hip), so the time of a fixed amount of multiply-add work per SIMD tells the residency.
// Written to explain the round-2 observation that the one-LDS-area type f kernel (36 KB per workgroup, 3.4 KB of scratch
// per lane, register budget for two waves per SIMD) ran with 1.4 resident waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int LDS_BYTES, int SCRATCH_BYTES, int WAVES>
__global__ void __launch_bounds__(128, WAVES) occ(uint64_t *out, uint32_t a, uint32_t b, int iters, int idx) {
  __shared__ uint32_t lds[LDS_BYTES / 4 + 1];
  volatile uint32_t priv[SCRATCH_BYTES / 4 + 1];
  priv[idx & (SCRATCH_BYTES / 4)] = a;                          // dynamic index: the array stays in scratch
  if (SCRATCH_BYTES) priv[(idx * 7) % (SCRATCH_BYTES / 4 + 1)] = b;
  lds[(threadIdx.x + idx) % (LDS_BYTES / 4 + 1)] = b;
  __syncthreads();
  uint64_t acc = threadIdx.x;
  uint32_t x = a + threadIdx.x + lds[(threadIdx.x * 3 + idx) % (LDS_BYTES / 4 + 1)], y = b ^ priv[(idx * 3) % (SCRATCH_BYTES / 4 + 1)];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 48; r++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
  }
  out[blockIdx.x * 128 + threadIdx.x] = acc + priv[idx % (SCRATCH_BYTES / 4 + 1)];
}

template <class K>
static void run(const char *name, K kern, int lds, int scratch, int waves) {
  uint64_t *d;
  // 16 waves' worth of work per SIMD however they are scheduled: 256 CUs x 4 SIMDs x 16 waves = 8192 workgroups of 2 waves
  const int blocks = 8192, iters = 2000;
  hipMalloc(&d, (size_t) blocks * 128 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(128), 0, 0, d, 12345u, 67890u, iters, rep);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double cyc = best * 1e-3 * 2.4e9, macs_per_simd = 16.0 * iters * 48;
  printf("%-10s LDS %6d B/workgroup  scratch %5d B/lane  budget %d waves/SIMD: %6.2f ms  %5.2f cycles per MAC per SIMD -> %s\n", name, lds,
         scratch, waves, best, cyc / macs_per_simd, cyc / macs_per_simd < 5.5 ? ">= 2 waves resident" : cyc / macs_per_simd < 8.0 ? "between 1 and 2" : "1 wave resident");
  hipFree(d);
}
#define RUN(L, S, W) run(#L "/" #S, occ<L, S, W>, L, S, W)
int main() {
  RUN(0, 0, 2);
  RUN(36864, 0, 2);
  RUN(39936, 0, 2);
  RUN(40960, 0, 2);
  RUN(45056, 0, 2);
  RUN(73728, 0, 1);
  RUN(0, 576, 2);
  RUN(0, 1024, 2);
  RUN(0, 2048, 2);
  RUN(0, 3456, 2);
  RUN(0, 8192, 2);
  RUN(36864, 576, 2);
  RUN(36864, 1024, 2);
  RUN(36864, 2048, 2);
  RUN(36864, 3456, 2);
  RUN(36864, 8192, 2);
  RUN(39936, 576, 2);
  RUN(39936, 3456, 2);
  RUN(0, 3456, 4);
  RUN(0, 0, 4);
  RUN(20480, 0, 4);
  RUN(20480, 1024, 4);
  return 0;
}
