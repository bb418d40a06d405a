// hostsim_shim.h -- TEST INFRASTRUCTURE: lets clang compile pbc_amd/csrc/*.cuh for the CPU so
// the exact kernel arithmetic (limb code, towers, Miller loops, final exponentiations) can be
// stepped through and checked against the golden vectors without a GPU.  One "lane" runs
// as an ordinary function call.  Never linked into libpbc_hip.so.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#define __device__
#define __global__
#define __constant__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
struct hostsim_dim3 { unsigned x, y, z; };
static hostsim_dim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0};
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r = {x, y, z, w}; return r; }
static inline int __all(int p) { return p; }      // one lane per call: the "wave" agrees with itself
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t s) {
  return (uint32_t) (((((uint64_t) hi) << 32) | lo) >> (s & 31));
}
// executed multiply-adds of everything run since the last reset (fp.cuh reports them through this hook)
static uint64_t hostsim_macs = 0;
#define PBC_COUNT_MACS(n) (hostsim_macs += (uint64_t) (n))
// the constant block the kernels read through the kernel argument segment on the GPU (fp.cuh, "KArgs"): here a
// host buffer that hostsim.cpp fills before each call
alignas(16) static uint8_t hostsim_kargs[4096];
static inline const uint8_t *pbc_kargs_base() { return hostsim_kargs; }      // KSEG = 4096 below the block's end
