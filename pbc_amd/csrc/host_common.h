// host_common.h -- what the translation units of libpbc_hip.so share: launch geometry, the dispatch over the built-in
// field widths, the constant block of a launch, and the functions that cross translation units.  The library is built
// from one .hip file per pairing family (pbc_hip_a.hip: types a, a1, e; pbc_hip_d.hip: types d, g; pbc_hip_f.hip: type f),
// one for the group operations (pbc_hip_group.hip) and pbc_hip.hip (the C-ABI, the host-buffer path, text formats,
// probes), compiled in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pbc_hip.h"
#include "fp.cuh"
#include "hostbn.h"
#include "pairing_a.cuh"
#include "pairing_al.cuh"
#include "pairing_d.cuh"
#include "pairing_f.cuh"
#include "pairing_e.cuh"
#include "group_al.cuh"
#include "group_l5.cuh"

using namespace pbc;

#include "host_params.h"

#define HIP_TRY(x)                                                                   \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) return fail("%s: %s", #x, hipGetErrorString(e_));          \
  } while (0)

// device allocation released on every return path
struct DevBuf {
  void *p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <class T> T *as() const { return static_cast<T *>(p); }
};
// the calling thread's current device is restored on every return path (callers such as torch keep their own)
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (dev >= 0 && dev != prev) (void) hipSetDevice(dev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
  ~DeviceGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

// ---------------------------------------------------------------------------------------
// launch geometry
// ---------------------------------------------------------------------------------------
constexpr int kBlock = 128;
#ifndef PBC_HIP_ZERO_COPY_DEFAULT
#define PBC_HIP_ZERO_COPY_DEFAULT 1
#endif
// Resident workgroups (the 5-word type f kernel).  The kernel is launched with at most as many workgroups as the chip
// holds at once (resident_grid below) and every workgroup walks the batch in strides of the grid: a lane runs its pairings
// one after the other.  With one workgroup per 128 units the dispatcher refills the CUs round by round, and with 36 KB of
// LDS per workgroup the rounds do not pack: a few workgroups find their LDS slot taken and wait for the NEXT round, so a
// 2^18 batch (two rounds of 1024 workgroups) takes three (tools/occ_schedule_probe.hip: mean residency 1.4 waves per SIMD; a
// single wave gets a multiply-add through only every 9.1 cycles, two share the pipe at 4.6).  All control flow is
// data-independent, so equal shares finish together.  Measured on the other kernels (types a, d, products,
// preprocessed pairings: 8 or more rounds, or LDS to spare): 3 - 4 % SLOWER than one workgroup per 128 units -- they keep
// the plain grid (profiles/r03_notes.md).
// Which 64-unit block a wave works on next: a fixed stride over the grid (ctr == null; iteration `it` of workgroup b's wave w
// takes block (b + it gridDim) 2 + w -- lane t of the workgroup works on unit (b + it gridDim) 128 + t), or the next value
// of a per-launch counter ("hip_dynamic 1": waves that finish early fetch more; control flow stays data-independent).
static __device__ __forceinline__ size_t pbc_unit_block(unsigned *ctr, size_t it) {
  unsigned v;
  if (ctr) {
    v = 0;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(ctr, 1u);
  } else {
    v = (unsigned) (((size_t) blockIdx.x + it * gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6));
  }
  return (size_t) (unsigned) __builtin_amdgcn_readfirstlane((int) v);
}
#define PBC_RESIDENT_LOOP(n, ctr) for (size_t it_ = 0, nvb_ = ((n) + 63) / 64, vb = pbc_unit_block(ctr, 0); vb < nvb_; vb = pbc_unit_block(ctr, ++it_))
#define PBC_UNIT_INDEX (vb * 64 + (threadIdx.x & 63))
static_assert(kBlock == D_LANES, "pairing_d.cuh sizes its LDS state for 128-lane workgroups");
#ifndef PBC_DF_WAVES
#define PBC_DF_WAVES 2
#endif
#ifndef PBC_A_WAVES
#define PBC_A_WAVES 2     // waves per SIMD the pairing kernels are register-budgeted for (measured: 1 -> 2 = +32 %)
#endif
#ifndef PBC_A1_WAVES
#define PBC_A1_WAVES 2    // 33-word fields.  Rounds 1-4 ran them at one wave per SIMD (512 registers; with the products in
#endif                    // registers two waves spilled: a1 158 k -> 119 k pairings/s).  On memory operands the product
                          // bodies need 121-248 registers and a second wave is what keeps the multiply-add pipe fed
                          // (profiles/r05_ab_wide.txt; 256-lane workgroups, pbc_hip_a.hip kWide)
#ifndef PBC_F_WAVES
#define PBC_F_WAVES PBC_DF_WAVES
#endif

// Type D / G kernels exist per (field width, d); N and DEG are compile-time inside the expression
#define PBC_DISPATCH_D(P_, ...)                                                        \
  switch ((P_)->nlimb * 8 + (P_)->deg) {                                               \
    case 5 * 8 + 3: { constexpr int N = 5, DEG = 3; __VA_ARGS__; } break;              \
    case 6 * 8 + 3: { constexpr int N = 6, DEG = 3; __VA_ARGS__; } break;              \
    case 7 * 8 + 3: { constexpr int N = 7, DEG = 3; __VA_ARGS__; } break;              \
    case 5 * 8 + 5: { constexpr int N = 5, DEG = 5; __VA_ARGS__; } break;              \
    default: return fail("internal: no type d/g kernel for %d-word fields, degree %d", (P_)->nlimb, (P_)->deg); \
  }

// type f kernels: 5-word (f.param) and 8-word (256-bit BN) fields
#define PBC_DISPATCH_F(nl, ...)                               \
  switch (nl) {                                               \
    case 5: { constexpr int N = 5; __VA_ARGS__; } break;      \
    case 8: { constexpr int N = 8; __VA_ARGS__; } break;      \
    default: return fail("internal: no type f kernel for %d-word fields", (int) (nl)); \
  }
// any built-in field width (PBC_FOR_EACH_N)
#define PBC_DISPATCH_N(nl, ...)                               \
  switch (nl) {                                               \
    case 5: { constexpr int N = 5; __VA_ARGS__; } break;             \
    case 6: { constexpr int N = 6; __VA_ARGS__; } break;             \
    case 7: { constexpr int N = 7; __VA_ARGS__; } break;             \
    case 8: { constexpr int N = 8; __VA_ARGS__; } break;             \
    case 16: { constexpr int N = 16; __VA_ARGS__; } break;           \
    case 33: { constexpr int N = 33; __VA_ARGS__; } break;           \
    default: return fail("internal: no kernel for %d-word fields", (int) (nl)); \
  }

// the constant block of an object, passed by value as the LAST argument of every kernel (fp.cuh, "KArgs")
template <int N>
static KArgs<N> kargs(const pbc_hip_pairing_s *P, bool for_pairing = false, bool no_fair = false) {
  KArgs<N> K;
  fill_kargs<N>(P, K, for_pairing);
  if (no_fair) { const uint32_t opt = 1u; memcpy(K.head + KOFF_OPT, &opt, 4); }    // this launch without time-sliced priorities
  return K;
}


// F = field policy of the G2 twist of an asymmetric type
#define PBC_DISPATCH_TWIST(P_, ...)                                                           \
  do {                                                                                        \
    if ((P_)->type == 'f') { PBC_DISPATCH_F((P_)->nlimb, { typedef Fq2Ops<N> F; __VA_ARGS__; }); } \
    else { PBC_DISPATCH_D(P_, { typedef FdOps<N, DEG> F; __VA_ARGS__; }); }                   \
  } while (0)

// ---------------------------------------------------------------------------------------
// functions that cross translation units
// ---------------------------------------------------------------------------------------
// pbc_hip.hip
int ensure_derived(pbc_hip_pairing_s *P, hipStream_t s);        // self-test + device-side derivation of an object's constants, once
unsigned resident_grid(const pbc_hip_pairing_s *P, const void *kernel, size_t n);
unsigned *unit_counter(pbc_hip_pairing_s *P, hipStream_t s);   // "hip_dynamic 1": a zeroed per-launch counter (enqueued on s); null otherwise
#define PBC_RGRID(...) resident_grid(P, reinterpret_cast<const void *>(&__VA_ARGS__), n)
void *pinned_dev_ptr(const void *host, size_t bytes, bool shared);
bool ranges_overlap(const void *a, size_t na, const void *b, size_t nb);
void *workspace_get(pbc_hip_pairing_s *P, hipStream_t s, size_t bytes);
void workspace_unpin(pbc_hip_pairing_s *P, hipStream_t s);
// A workspace that belongs to one stream of a device context of the host-buffer path: grown on demand, freed with the context.
struct OwnWs { void **p; size_t *cap; };
void *own_workspace(const OwnWs &o, hipStream_t s, size_t bytes);
void *object_scratch(pbc_hip_pairing_s *P, const void *key, size_t bytes, bool *fresh);    // small per-(device, key) buffers the object keeps (pbc_hip.hip)
// The workspace of one product launch: the caller's own (host-buffer path) or the object's table entry for (device,
// stream), which stays pinned until the launch's kernels are enqueued (the destructor unpins).
struct ProdWs {
  pbc_hip_pairing_s *P;
  hipStream_t s;
  const OwnWs *own;
  bool pinned = false;
  ProdWs(pbc_hip_pairing_s *P_, hipStream_t s_, const OwnWs *own_) : P(P_), s(s_), own(own_) {}
  ProdWs(const ProdWs &) = delete;
  ProdWs &operator=(const ProdWs &) = delete;
  ~ProdWs() { if (pinned) workspace_unpin(P, s); }
  void *get(size_t bytes) {
    if (own) return own_workspace(*own, s, bytes);
    void *w = workspace_get(P, s, bytes);      // pins the entry and takes its issue lock (pbc_hip.hip WsEnt)
    if (w && pinned) workspace_unpin(P, s);    // a second request of the same call: one pin, one lock level
    pinned = pinned || w != nullptr;
    return w;
  }
};
// The host-buffer path (pbc_hip.hip): chunks over the device set, page-locked buffers in place, anything else staged
// through per-device chunk buffers that the object keeps.  `launch` enqueues one chunk (device pointers) on a stream.
typedef std::function<int(void *d_out, const void *d_a, const void *d_b, size_t m, hipStream_t s, const OwnWs *own)> ChunkLaunch;
int run_host_generic(pbc_hip_pairing_s *P, uint8_t *out, size_t ut, const uint8_t *a, size_t u1, const uint8_t *b, size_t u2,
                     size_t n, const ChunkLaunch &launch, bool zero_copy_ok);
// Per-family launchers: k-term products (k = 1: single pairings) of n units on stream s; constants already derived.
int launch_a(pbc_hip_pairing_s *P, void *d_gt, const void *d_g1, const void *d_g2, size_t n, int k, hipStream_t s, ProdWs &W);   // pbc_hip_a.hip: a, a1, e
int launch_d(pbc_hip_pairing_s *P, void *d_gt, const void *d_g1, const void *d_g2, size_t n, int k, hipStream_t s, ProdWs &W);   // pbc_hip_d.hip: d, g
int launch_f(pbc_hip_pairing_s *P, void *d_gt, const void *d_g1, const void *d_g2, size_t n, int k, hipStream_t s, ProdWs &W);   // pbc_hip_f.hip
// one-time derivations on the device (single-lane kernels; results land in the object)
int derive_d(pbc_hip_pairing_s *P, hipStream_t s);
int derive_e(pbc_hip_pairing_s *P, hipStream_t s);
int derive_f(pbc_hip_pairing_s *P, hipStream_t s);
// preprocessed pairings: table derivation (one lane) and application (one second argument per lane)
struct pbc_hip_pp_s {
  pbc_hip_pairing_s *P;
  uint32_t *tab;      // device: type a [exp2 + 1][3][16] words; types d / g [steps][3][ND] words
  uint32_t *valid;    // device flag: first argument was a finite curve point
};
void pp_init_launch_a(pbc_hip_pairing_s *P, pbc_hip_pp_s *pp, const uint8_t *d_g1, bool a1);
int pp_init_launch_d(pbc_hip_pairing_s *P, pbc_hip_pp_s *pp, const uint8_t *d_g1);
int pp_apply_launch_a(pbc_hip_pp_s *pp, void *d_gt, const void *d_g2, size_t n, hipStream_t s);
int pp_apply_launch_d(pbc_hip_pp_s *pp, void *d_gt, const void *d_g2, size_t n, hipStream_t s);
// type f diagnostics (pbc_hip_diag_stage)
int diag_f_miller(pbc_hip_pairing_s *P, void *dt, const void *d1, const void *d2, size_t n);
int diag_f_op(pbc_hip_pairing_s *P, int stage, void *dt, const void *d1, const void *d2, size_t n);
