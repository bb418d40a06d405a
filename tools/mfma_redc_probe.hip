// mfma_redc_probe.hip -- VERDICT r3 item 7, gated experiment: could the Montgomery reduction of the 512-bit product run on
// the int8 MFMA pipe?  Half of a product's multiply-adds are m q and T (-1/q) with wave-uniform constants, i.e. the dense
// contraction Toeplitz(q) [128 x 64 bytes] x M [64 bytes x 64 lanes].  This probe measures the three costs the idea stands
// on, separately, in cycles per wave and REDC, at one and two waves per SIMD:
//   (1) valu  : the reduction as the kernels run it (fp.cuh wide_reduce shape: 18 x 18 multiply-adds on 29-bit limbs);
//   (2) mfma  : 24 x v_mfma_i32_32x32x32_i8 -- the 16 of m q (128 x 64 x 64) and the 8 of the lower half of T (-1/q);
//   (3) glue  : what the VALU still has to do per lane for an MFMA reduction: 64 + 128 int32 column sums at radix 2^8
//               packed four at a time into 64-bit words, the carry chain over those words, and the way back to 18 limbs
//               (operand shuffles -- v_permlane32_swap for the half-wave split of the B operand and the D results -- are
//               NOT included: the estimate is a lower bound).
// The MFMA reduction can only pay if  glue + max(0, mfma - overlap)  is well below  valu ; the gate was 1.25 x on the
// whole product (a b on the VALU stays: 324 multiply-adds).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_redc_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int WHAT>
__global__ void __launch_bounds__(64) probe(uint64_t *out, const uint32_t *in, int iters) {
  const int lane = threadIdx.x;
  uint32_t x[18], q[18];
  for (int i = 0; i < 18; i++) { x[i] = in[lane * 18 + i] & 0x1fffffffu; q[i] = in[(64 + i) * 7 % 1024] & 0x1fffffffu; }
  uint64_t W[35];
  for (int i = 0; i < 35; i++) W[i] = ((uint64_t) in[(lane + i) & 1023] << 20) | in[(lane * 3 + i) & 1023];
  v4i A = {(int) in[lane], (int) in[lane + 64], (int) in[lane + 128], (int) in[lane + 192]}, B = A;
  v16i acc[8];
  for (int t = 0; t < 8; t++) for (int k = 0; k < 16; k++) acc[t][k] = 0;
  uint64_t sink = 0;
  const uint64_t t0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    if constexpr (WHAT == 0) {                       // the VALU reduction: columns of W plus m q, m from the low columns
      uint32_t m[18];
      uint64_t a = 0;
#pragma unroll
      for (int k = 0; k < 18; k++) {
        a += W[k];
#pragma unroll
        for (int i = 0; i < k; i++) a += (uint64_t) m[i] * q[k - i];
        m[k] = ((uint32_t) a * 0x0badcafu) & 0x1fffffffu;
        a += (uint64_t) m[k] * q[0];
        a >>= 29;
      }
#pragma unroll
      for (int k = 18; k < 36; k++) {
        if (k < 35) a += W[k];
#pragma unroll
        for (int i = k - 17; i < 18; i++) a += (uint64_t) m[i] * q[k - i];
        x[k - 18] = (uint32_t) a & 0x1fffffffu;
        a >>= 29;
      }
#pragma unroll
      for (int i = 0; i < 18; i++) W[i] += x[i];
    } else if constexpr (WHAT == 1) {                // 24 MFMAs: 8 output tiles x 2 K-steps (m q) + 4 x 2 (T (-1/q), lower half)
#pragma unroll
      for (int t = 0; t < 8; t++) {
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(B, A, acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; t++) {
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, A, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(B, B, acc[t], 0, 0, 0);
      }
      A[0] ^= acc[0][0] & 1;
    } else {                                         // glue: 192 radix-2^8 column sums -> 48 64-bit words -> carries -> 18 limbs
      uint32_t col[192];
#pragma unroll
      for (int i = 0; i < 192; i++) col[i] = (uint32_t) (W[i % 35] >> (i / 35)) & 0x7fffffu;      // stand-ins for the MFMA results
      uint64_t w[48];
#pragma unroll
      for (int i = 0; i < 48; i++)
        w[i] = (uint64_t) col[4 * i] + ((uint64_t) col[4 * i + 1] << 8) + ((uint64_t) col[4 * i + 2] << 16) + ((uint64_t) col[4 * i + 3] << 24);
      uint64_t c = 0;
      uint32_t words[48];
#pragma unroll
      for (int i = 0; i < 48; i++) { c += w[i]; words[i] = (uint32_t) c; c >>= 32; }
#pragma unroll
      for (int i = 0; i < 18; i++) {                 // the upper 512 bits back to 29-bit limbs
        const int bit = 29 * i, j = 32 + (bit >> 5), sh = bit & 31;
        const uint64_t pair = ((uint64_t) words[j + 1 < 48 ? j + 1 : 47] << 32) | words[j < 48 ? j : 47];
        x[i] = (uint32_t) (pair >> sh) & 0x1fffffffu;
      }
#pragma unroll
      for (int i = 0; i < 18; i++) W[i] += x[i];
    }
  }
  const uint64_t t1 = wall_clock64();
  for (int i = 0; i < 18; i++) sink += x[i] + W[i];
  for (int t = 0; t < 8; t++) sink += (uint64_t) acc[t][3];
  if (lane == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = sink; }
}

int main() {
  const int iters = 2000;
  uint32_t *in;
  uint64_t *out;
  hipMalloc(&in, 4096 * 4);
  hipMalloc(&out, 8192 * 16);
  uint32_t h[4096];
  for (int i = 0; i < 4096; i++) h[i] = 2654435761u * (i + 1) + 12345u;
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double wall_hz = 100e6;                      // wall_clock64 ticks at 100 MHz on gfx950
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const char *name[3] = {"valu REDC (324 multiply-adds, 29-bit limbs)", "mfma: 24 x v_mfma_i32_32x32x32_i8", "glue: 192 columns -> carries -> 18 limbs"};
  for (int waves = 1; waves <= 2; waves++)
    for (int what = 0; what < 3; what++) {
      const int grid = prop.multiProcessorCount * 4 * waves;      // `waves` single-wave workgroups per SIMD
      uint64_t *ho = new uint64_t[2 * grid];
      for (int rep = 0; rep < 2; rep++) {
        if (what == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(64), 0, 0, out, in, iters);
        else if (what == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 0, 0, out, in, iters);
        else hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(64), 0, 0, out, in, iters);
        hipDeviceSynchronize();
      }
      hipMemcpy(ho, out, 16 * grid, hipMemcpyDeviceToHost);
      double sum = 0;
      for (int b = 0; b < grid; b++) sum += (double) ho[2 * b];
      const double ticks = sum / grid / iters, cycles = ticks / wall_hz * clk_khz * 1e3;
      printf("%d wave(s)/SIMD  %-46s %8.1f cycles per wave and REDC (shader clock %d MHz)\n", waves, name[what], cycles, clk_khz / 1000);
      delete[] ho;
    }
  return 0;
}
