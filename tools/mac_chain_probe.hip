// Standalone probe (hipcc --offload-arch=gfx950 -O2 tools/mac_chain_probe.hip -o tools/exp/mac_chain): cycles per
// v_mad_u64_u32 when the multiply-adds form C independent accumulator chains, at W waves per SIMD.
// The multiplier's column sums are ONE chain (acc += x_i y_j, each instruction waits for the one before).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int C>
__global__ void __launch_bounds__(64) chains(uint64_t *out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc[C];
  uint32_t x[4], y[4];
  for (int i = 0; i < 4; i++) { x[i] = a + threadIdx.x * (i + 1); y[i] = b ^ (threadIdx.x << i); }
  for (int c = 0; c < C; c++) acc[c] = threadIdx.x + c;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 48 / C; r++)
#pragma unroll
      for (int c = 0; c < C; c++)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x[(r + c) & 3]), "v"(y[(r * 3 + c) & 3]) : "vcc");
  }
  uint64_t s = 0;
  for (int c = 0; c < C; c++) s += acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
// the same chain with independent cheap instructions in between (the shifts / masks / adds of real code)
template <int C, int F>
__global__ void __launch_bounds__(64) chains_mixed(uint64_t *out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc[C];
  uint32_t x[4], y[4], z[4];
  for (int i = 0; i < 4; i++) { x[i] = a + threadIdx.x * (i + 1); y[i] = b ^ (threadIdx.x << i); z[i] = i; }
  for (int c = 0; c < C; c++) acc[c] = threadIdx.x + c;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 48 / C; r++)
#pragma unroll
      for (int c = 0; c < C; c++) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x[(r + c) & 3]), "v"(y[(r * 3 + c) & 3]) : "vcc");
#pragma unroll
        for (int f = 0; f < F; f++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(z[(r + f) & 3]) : "v"(x[f & 3]));
      }
  }
  uint64_t s = z[0] + z[1] + z[2] + z[3];
  for (int c = 0; c < C; c++) s += acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class K>
static void run(const char *name, K kern, int waves_per_simd, int macs_per_iter, int extra) {
  uint64_t *d;
  const int blocks = 256 * 4 * waves_per_simd, iters = 20000;
  hipMalloc(&d, (size_t) blocks * 64 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 12345u, 67890u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) {
      const double cyc = ms * 1e-3 * 2.4e9, per_simd = (double) waves_per_simd * iters * macs_per_iter;
      printf("%-28s %d waves/SIMD: %6.2f cycles per MAC per SIMD  (%.2f ms, +%d adds per MAC)\n", name, waves_per_simd, cyc / per_simd, ms, extra);
    }
  }
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) {
    run("1 chain", chains<1>, w, 48, 0);
    run("2 chains", chains<2>, w, 48, 0);
    run("3 chains", chains<3>, w, 48, 0);
    run("4 chains", chains<4>, w, 48, 0);
    run("8 chains", chains<8>, w, 48, 0);
    run("1 chain + 1 add each", chains_mixed<1, 1>, w, 48, 1);
    run("1 chain + 2 adds each", chains_mixed<1, 2>, w, 48, 2);
    run("2 chains + 1 add each", chains_mixed<2, 1>, w, 48, 1);
  }
  return 0;
}
