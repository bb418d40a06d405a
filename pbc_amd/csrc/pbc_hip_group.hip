// pbc_hip_group.hip -- kernels and C-ABI entry points of the group operations next to the pairing: libpbc_hip.so; see host_common.h
#include "host_common.h"

// ---- group operations (one element per lane) -------------------------------------------------
template <int N>
__global__ void __launch_bounds__(kBlock, 2) g_mul_kernel(uint8_t *out, const uint8_t *in, const uint8_t *z,
                                                           int zlen, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = 2 * fpk<N>().fbytes;
  g_mul_lane<N>(out + idx * L, in + idx * L, z + idx * zlen, zlen);
}
// element_to_bytes_compressed / _x_only and element_from_bytes_compressed / _x_only on E(F_q): one point per lane
// (dir 0 / 2 and 1 / 3)
template <int N>
__global__ void __launch_bounds__(kBlock, 2) g_compress_kernel(int dir, uint8_t *out, const uint8_t *in, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t fb = fpk<N>().fbytes;
  if (dir == 0) g_compress_lane<N>(out + idx * (fb + 1), in + idx * 2 * fb);
  else if (dir == 1) g_decompress_lane<N>(out + idx * 2 * fb, in + idx * (fb + 1));
  else if (dir == 2) g_to_x_only_lane<N>(out + idx * fb, in + idx * 2 * fb);
  else g_decompress_lane<N>(out + idx * 2 * fb, in + idx * fb, true);
}
// element_mul_zn on the twists: G2 of types d / g (over F_q^d) and f (over F_q^2)
template <int N, int DEG>
__global__ void __launch_bounds__(kBlock, 2) d_g2_mul_kernel(uint8_t *out, const uint8_t *in, const uint8_t *z,
                                                              int zlen, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = 2 * DEG * fpk<N>().fbytes;
  ec_mul_lane<FdOps<N, DEG>>(out + idx * L, in + idx * L, z + idx * zlen, zlen);
}
template <int N>
__global__ void __launch_bounds__(kBlock, 2) f_g2_mul_kernel(uint8_t *out, const uint8_t *in, const uint8_t *z,
                                                              int zlen, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = 4 * fpk<N>().fbytes;
  ec_mul_lane<Fq2Ops<N>>(out + idx * L, in + idx * L, z + idx * zlen, zlen);
}
template <int N>
__global__ void __launch_bounds__(kBlock, 2) g_from_hash_kernel(uint8_t *out, const uint8_t *data, int hlen, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  size_t ld = idx < n ? idx : n - 1;   // whole waves stay in the retry loop together
  const int L = 2 * (int) fpk<N>().fbytes;
  __attribute__((aligned(4))) uint8_t o[8 * N];
  g_from_hash_lane<N>(o, data + ld * hlen, hlen);
  if (idx < n)
    for (int i = 0; i < L; i++) out[idx * L + i] = o[i];
}
// element_from_hash / element_to_bytes_compressed / element_from_bytes_compressed on the G2 twists (types d, g, f):
// F is the field policy of the twist (FdOps / Fq2Ops).  what 0: digests of `aux` bytes -> points; 1: points ->
// x || s; 2: x || s -> points; 3: points -> x; 4: x -> points
template <class F>
__global__ void __launch_bounds__(kBlock, 2) g2_point_kernel(int what, uint8_t *out, const uint8_t *in, int aux, size_t n, KArgs<F::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  size_t ld = idx < n ? idx : n - 1;   // whole waves stay in the retry loop together
  const size_t fb = (size_t) F::bytes();
  const size_t li = what == 0 ? (size_t) aux : (what == 1 || what == 3) ? 2 * fb : what == 2 ? fb + 1 : fb;
  const size_t lo = what == 1 ? fb + 1 : what == 3 ? fb : 2 * fb;
  __attribute__((aligned(4))) uint8_t o[8 * F::WORDS];
  if (what == 0) g2_from_hash_lane<F>(o, in + ld * li, aux);
  else if (what == 1) g2_compress_lane<F>(o, in + ld * li);
  else if (what == 2) g2_decompress_lane<F>(o, in + ld * li);
  else if (what == 3) { for (size_t i = 0; i < fb; i++) o[i] = in[ld * li + i]; }
  else g2_from_x_lane<F>(o, in + ld * li);
  if (idx < n)
    for (size_t i = 0; i < lo; i++) out[idx * lo + i] = o[i];
}
template <class F>
__global__ void ext_ts_init_kernel(uint32_t *out, KArgs<F::NW> ka) {
  if (threadIdx.x || blockIdx.x) return;
  ext_ts_init<F>(out);
}
// one lane: z^t' for the Tonelli-Shanks square roots of element_from_hash (fields with q = 1 mod 4)
struct TsRaw { uint32_t t[34], half[34]; int tbits, halfbits; };
template <int N>
__global__ void ts_init_kernel(uint32_t *out, TsRaw raw, KArgs<N> ka) {
  if (threadIdx.x || blockIdx.x) return;
  fp_ts_init<N>(out, raw.t, raw.tbits, raw.half, raw.halfbits);
}
// op 0: out = a * b in GT;  op 1: out = a ^ z;  op 2: out = finalpow(a) (the final exponentiation alone)
template <int N>
__global__ void __launch_bounds__(kBlock, 2) gt_op_kernel(int type, int op, uint8_t *out, const uint8_t *a,
                                                           const uint8_t *b, int lenT, int zlen, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  uint8_t *o = out + idx * lenT;
  const uint8_t *x = a + idx * lenT;
  if constexpr (N == 16 || N == 33) {
    if (type == 'e') {                 // GT = F_q (pairing_GT_init(pairing, p->Fq), e_param.c:863)
      fp<N> u, v;
      if (op == 2) { e_finalpow_lane<N>(o, x); return; }
      fp_load_be<N>(u, x);
      if (op == 0) {
        fp_load_be<N>(v, b + idx * lenT);
        fp_mul<N>(u, u, v);
      } else {
        fp<N> acc, t;
        fp_set<N>(acc, fpk<N>().one);
        const uint8_t *z = b + idx * zlen;
        for (int i = 8 * zlen - 1; i >= 0; i--) {
          fp_sqr<N>(acc, acc);
          fp_mul<N>(t, acc, u);
          fp_cmov<N>(acc, t, zr_bit(z, zlen, i) != 0);
        }
        u = acc;
      }
      fp_store_be<N>(o, u);
      return;
    }
  }
  if constexpr (N == 16 || N == 33) {
    if (op == 0) a_gt_mul_lane<N>(o, x, b + idx * lenT); else if (op == 1) a_gt_pow_lane<N>(o, x, b + idx * zlen, zlen); else a_finalpow_lane<N>(o, x);
  } else {
    if (type == 'd') {
      if constexpr (N <= ND_MAX) {
        if (op == 0) d_gt_mul_lane<N, 3>(o, x, b + idx * lenT); else if (op == 1) d_gt_pow_lane<N, 3>(o, x, b + idx * zlen, zlen); else d_finalpow_lane<N, 3>(o, x);
      }
    } else if constexpr (N == 5 || N == 8) {
      if (type == 'g') {
        if constexpr (N == 5) {
          if (op == 0) d_gt_mul_lane<N, 5>(o, x, b + idx * lenT); else if (op == 1) d_gt_pow_lane<N, 5>(o, x, b + idx * zlen, zlen); else d_finalpow_lane<N, 5>(o, x);
        }
      } else {
        if (op == 0) f_gt_mul_lane<N>(o, x, b + idx * lenT); else if (op == 1) f_gt_pow_lane<N>(o, x, b + idx * zlen, zlen); else f_finalpow_lane<N>(o, x);
      }
    }
  }
}

// ---- group operations ------------------------------------------------------------------------
extern "C" int pbc_hip_pairing_length_in_bytes_Zr(const pbc_hip_pairing_t *p) { return p->len_zr; }

// three device buffers in, one out: shared host path for the group-operation entry points
static int run_group(pbc_hip_pairing_s *P, int what, int group, uint8_t *out, const uint8_t *a, const uint8_t *b,
                     size_t n) {
  if (P->device < 0) return fail("no HIP device: libpbc_hip has no CPU fallback");
  if (!n) return 0;
  size_t la, lb, lo;
  if (what == 0) {                     // G mul_zn
    if (group != 1 && group != 2) return fail("group must be 1 or 2");
    la = lo = (size_t) (group == 1 ? P->len1 : P->len2);
    lb = (size_t) P->len_zr;
  } else if (what == 1) {              // GT mul
    la = lb = lo = (size_t) P->lenT;
  } else if (what == 2) {              // GT pow
    la = lo = (size_t) P->lenT;
    lb = (size_t) P->len_zr;
  } else {                             // finalpow: one operand
    la = lo = (size_t) P->lenT;
    lb = 0;
  }
  DevBuf ba, bb, bo;
  DeviceGuard guard(P->device);
  HIP_TRY(ba.alloc(n * la));
  HIP_TRY(bb.alloc(n * lb));
  HIP_TRY(bo.alloc(n * lo));
  void *da = ba.p, *db = bb.p, *d_o = bo.p;
  HIP_TRY(hipMemcpy(da, a, n * la, hipMemcpyHostToDevice));
  if (lb) HIP_TRY(hipMemcpy(db, b, n * lb, hipMemcpyHostToDevice));
  if (ensure_derived(P, 0)) return 1;
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  if (what == 0 && group == 2 && (P->type == 'd' || P->type == 'g')) {
    PBC_DISPATCH_D(P, hipLaunchKernelGGL((d_g2_mul_kernel<N, DEG>), dim3(grid), dim3(kBlock), 0, 0, (uint8_t *) d_o,
                                         (const uint8_t *) da, (const uint8_t *) db, P->len_zr, n, kargs<N>(P)));
  } else if (what == 0 && group == 2 && P->type == 'f') {
    PBC_DISPATCH_F(P->nlimb, hipLaunchKernelGGL(f_g2_mul_kernel<N>, dim3(grid), dim3(kBlock), 0, 0, (uint8_t *) d_o, (const uint8_t *) da,
                       (const uint8_t *) db, P->len_zr, n, kargs<N>(P)));
  } else if (what == 0) {              // E(F_q): G1, and G2 of the symmetric types
    PBC_DISPATCH_N(P->nlimb, hipLaunchKernelGGL(g_mul_kernel<N>, dim3(grid), dim3(kBlock), 0, 0, (uint8_t *) d_o,
                                                (const uint8_t *) da, (const uint8_t *) db, P->len_zr, n, kargs<N>(P)));
  } else {
    PBC_DISPATCH_N(P->nlimb, hipLaunchKernelGGL(gt_op_kernel<N>, dim3(grid), dim3(kBlock), 0, 0, P->type, what - 1,
                                                (uint8_t *) d_o, (const uint8_t *) da, (const uint8_t *) db, P->lenT,
                                                P->len_zr, n, kargs<N>(P)));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, d_o, n * lo, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int pbc_hip_element_mul_zn_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *in,
                                            const uint8_t *zr, size_t n) {
  if (!P) return fail("null pairing");
  return run_group(P, 0, group, out, in, zr, n);
}
extern "C" int pbc_hip_element_mul_GT_batch(pbc_hip_pairing_t *P, uint8_t *out, const uint8_t *a, const uint8_t *b,
                                            size_t n) {
  if (!P) return fail("null pairing");
  return run_group(P, 1, 0, out, a, b, n);
}
extern "C" int pbc_hip_element_pow_zn_GT_batch(pbc_hip_pairing_t *P, uint8_t *out, const uint8_t *a,
                                               const uint8_t *zr, size_t n) {
  if (!P) return fail("null pairing");
  return run_group(P, 2, 0, out, a, zr, n);
}
extern "C" int pbc_hip_finalpow_batch(pbc_hip_pairing_t *P, uint8_t *out, const uint8_t *in, size_t n) {
  if (!P) return fail("null pairing");
  return run_group(P, 3, 0, out, in, nullptr, n);
}

// first use of a square root in a field with q = 1 mod 4: derive the non-residue power z^t of the
// Tonelli-Shanks tail on the device (single lane)
static int ensure_sqrt_constants(pbc_hip_pairing_s *P) {
  if (!P->hash.ts_ready) {
    if (ensure_derived(P, 0)) return 1;
    DevBuf bc;
    TsRaw raw;
    memcpy(raw.t, P->hash.ts_t, sizeof raw.t);
    memcpy(raw.half, P->hash.half, sizeof raw.half);
    raw.tbits = P->hash.ts_tbits;
    raw.halfbits = P->hash.halfbits;
    HIP_TRY(bc.alloc(sizeof P->hash.ts_c));
    uint32_t *dc = bc.as<uint32_t>();
    HIP_TRY(hipMemset(dc, 0, sizeof P->hash.ts_c));
    PBC_DISPATCH_N(P->nlimb, hipLaunchKernelGGL(ts_init_kernel<N>, dim3(1), dim3(64), 0, 0, dc, raw, kargs<N>(P)));
    HIP_TRY(hipMemcpy(P->hash.ts_c, dc, sizeof P->hash.ts_c, hipMemcpyDeviceToHost));
    P->hash.ts_ready = true;
  }
  return 0;
}
// F = field policy of the G2 twist of an asymmetric type
// z^T for the square roots in the twist's field, once per parameter set
static int ensure_ext_sqrt(pbc_hip_pairing_s *P) {
  if (!P->xs_ready) {
    if (ensure_derived(P, 0)) return 1;
    DevBuf bc;
    HIP_TRY(bc.alloc(sizeof P->xs.c));
    uint32_t *dc = bc.as<uint32_t>();
    HIP_TRY(hipMemset(dc, 0, sizeof P->xs.c));
    PBC_DISPATCH_TWIST(P, hipLaunchKernelGGL(ext_ts_init_kernel<F>, dim3(1), dim3(64), 0, 0, dc, kargs<F::NW>(P)));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(P->xs.c, dc, sizeof P->xs.c, hipMemcpyDeviceToHost));
    P->xs_ready = true;
  }
  return 0;
}
// what 0: element_from_hash (li = hlen), 1 / 2: to / from_bytes_compressed, 3 / 4: to / from_bytes_x_only -- on the G2 twist
static int run_twist_points(pbc_hip_pairing_s *P, int what, uint8_t *out, const uint8_t *in, int hlen, size_t n) {
  const size_t lp = (size_t) P->len2, lc = lp / 2 + 1, lx = lp / 2;
  const size_t li = what == 0 ? (size_t) hlen : (what == 1 || what == 3) ? lp : what == 2 ? lc : lx;
  const size_t lo = what == 1 ? lc : what == 3 ? lx : lp;
  DevBuf bi, bo;
  DeviceGuard guard(P->device);
  if (ensure_ext_sqrt(P)) return 1;
  if (P->type == 'f' && ensure_sqrt_constants(P)) return 1;   // fq_sqrt works through square roots in F_q
  HIP_TRY(bi.alloc(n * li));
  HIP_TRY(bo.alloc(n * lo));
  void *di = bi.p, *d_o = bo.p;
  HIP_TRY(hipMemcpy(di, in, n * li, hipMemcpyHostToDevice));
  if (ensure_derived(P, 0)) return 1;
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  PBC_DISPATCH_TWIST(P, hipLaunchKernelGGL(g2_point_kernel<F>, dim3(grid), dim3(kBlock), 0, 0, what, (uint8_t *) d_o,
                                           (const uint8_t *) di, hlen, n, kargs<F::NW>(P)));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, d_o, n * lo, hipMemcpyDeviceToHost));
  return 0;
}
// dir 0: x||y -> x||s;  dir 1: x||s -> x||y;  dir 2: x||y -> x;  dir 3: x -> x||y
static int run_compress(pbc_hip_pairing_s *P, int dir, int group, uint8_t *out, const uint8_t *in, size_t n) {
  if (!P) return fail("null pairing");
  if (P->device < 0) return fail("no HIP device: libpbc_hip has no CPU fallback");
  const bool symmetric = P->type == 'a' || P->type == '1' || P->type == 'e';
  if (group == 2 && !symmetric) return n ? run_twist_points(P, dir + 1, out, in, 0, n) : 0;
  if (group != 1 && group != 2) return fail("group must be 1 or 2");
  if (!n) return 0;
  const size_t lp = (size_t) P->len1, lc = (size_t) P->len_fq + (dir < 2 ? 1 : 0);
  const size_t li = (dir & 1) == 0 ? lp : lc, lo = (dir & 1) == 0 ? lc : lp;
  DevBuf bi, bo;
  DeviceGuard guard(P->device);
  if (ensure_sqrt_constants(P)) return 1;
  HIP_TRY(bi.alloc(n * li));
  HIP_TRY(bo.alloc(n * lo));
  void *di = bi.p, *d_o = bo.p;
  HIP_TRY(hipMemcpy(di, in, n * li, hipMemcpyHostToDevice));
  if (ensure_derived(P, 0)) return 1;
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  PBC_DISPATCH_N(P->nlimb, hipLaunchKernelGGL(g_compress_kernel<N>, dim3(grid), dim3(kBlock), 0, 0, dir, (uint8_t *) d_o,
                                              (const uint8_t *) di, n, kargs<N>(P)));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, d_o, n * lo, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int pbc_hip_element_to_bytes_compressed_batch(pbc_hip_pairing_t *P, int group, uint8_t *out,
                                                         const uint8_t *in, size_t n) {
  return run_compress(P, 0, group, out, in, n);
}
extern "C" int pbc_hip_element_from_bytes_compressed_batch(pbc_hip_pairing_t *P, int group, uint8_t *out,
                                                           const uint8_t *in, size_t n) {
  return run_compress(P, 1, group, out, in, n);
}
extern "C" int pbc_hip_pairing_length_in_bytes_compressed_G1(const pbc_hip_pairing_t *p) { return p->len_fq + 1; }
extern "C" int pbc_hip_pairing_length_in_bytes_compressed_G2(const pbc_hip_pairing_t *p) { return p->len2 / 2 + 1; }
extern "C" int pbc_hip_element_to_bytes_x_only_batch(pbc_hip_pairing_t *P, int group, uint8_t *out,
                                                     const uint8_t *in, size_t n) {
  return run_compress(P, 2, group, out, in, n);
}
extern "C" int pbc_hip_element_from_bytes_x_only_batch(pbc_hip_pairing_t *P, int group, uint8_t *out,
                                                       const uint8_t *in, size_t n) {
  return run_compress(P, 3, group, out, in, n);
}
extern "C" int pbc_hip_pairing_length_in_bytes_x_only_G1(const pbc_hip_pairing_t *p) { return p->len_fq; }
extern "C" int pbc_hip_pairing_length_in_bytes_x_only_G2(const pbc_hip_pairing_t *p) { return p->len2 / 2; }

extern "C" int pbc_hip_element_from_hash_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *data,
                                               int hlen, size_t n) {
  if (!P) return fail("null pairing");
  if (P->device < 0) return fail("no HIP device: libpbc_hip has no CPU fallback");
  const bool symmetric = P->type == 'a' || P->type == '1' || P->type == 'e';
  if (group != 1 && group != 2) return fail("group must be 1 or 2");
  if (hlen < 1) return fail("hlen must be >= 1");
  if (!n) return 0;
  if (group == 2 && !symmetric) {
    if (P->type == 'f' && hlen < 2) return fail("type f G2: hlen must be >= 2 (fq_from_hash halves the digest)");
    return run_twist_points(P, 0, out, data, hlen, n);
  }
  DevBuf bd, bo;
  DeviceGuard guard(P->device);
  if (ensure_sqrt_constants(P)) return 1;
  HIP_TRY(bd.alloc(n * (size_t) hlen));
  HIP_TRY(bo.alloc(n * (size_t) P->len1));
  void *dd = bd.p, *d_o = bo.p;
  HIP_TRY(hipMemcpy(dd, data, n * (size_t) hlen, hipMemcpyHostToDevice));
  if (ensure_derived(P, 0)) return 1;
  unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  PBC_DISPATCH_N(P->nlimb, hipLaunchKernelGGL(g_from_hash_kernel<N>, dim3(grid), dim3(kBlock), 0, 0, (uint8_t *) d_o,
                                              (const uint8_t *) dd, hlen, n, kargs<N>(P)));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, d_o, n * (size_t) P->len1, hipMemcpyDeviceToHost));
  return 0;
}
