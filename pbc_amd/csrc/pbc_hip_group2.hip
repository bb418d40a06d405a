// pbc_hip_group2.hip -- kernels and C-ABI entry points of the group law, Z_r arithmetic and the multi-exponentiations
// (round 5; group_more.cuh): libpbc_hip.so; see host_common.h
#include "host_common.h"
#include "group_more.cuh"

// element_add / element_sub / element_neg / element_double on G1 / G2: one record (pair) per lane
template <class F>
__global__ void __launch_bounds__(kBlock, 2) ec_affine_op_kernel(int op, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n, KArgs<F::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = 2 * (size_t) F::bytes();
  ec_affine_op_lane<F>(op, out + idx * L, a + idx * L, b ? b + idx * L : a + idx * L);
}
// element_pow2_zn / element_pow3_zn on G1 / G2 and on GT
// (fast pass: one doubling + one incomplete addition per bit, lanes it cannot finish are flagged; complete pass: the
// flagged lanes -- flags == null: every lane)
template <class F>
__global__ void __launch_bounds__(kBlock, 2) ec_multi_mul_fast_kernel(uint8_t *out, MultiArgs M, int k, int zlen, uint8_t *flags, size_t n, KArgs<F::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  flags[idx] = ec_multi_mul_fast_lane<F>(out + idx * 2 * (size_t) F::bytes(), M, idx, k, zlen) ? 0 : 1;
}
template <class F>
__global__ void __launch_bounds__(kBlock, 2) ec_multi_mul_kernel(uint8_t *out, MultiArgs M, int k, int zlen, const uint8_t *flags, size_t n, KArgs<F::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  if (flags && !flags[idx]) return;
  ec_multi_mul_lane<F>(out + idx * 2 * (size_t) F::bytes(), M, idx, k, zlen);
}
// Type a, 512-bit field, G1 / G2: the joint signed-window ladder on the limb-form arithmetic (group_al.cuh gmulk_lane), resident
// workgroups as the single-base ladder; lanes it reports are left to ec_multi_mul_kernel
template <int N, int KB>
__global__ void __launch_bounds__(kBlock, PBC_A_WAVES) al_gmulk_kernel(uint8_t *out, MultiArgs M, int zlen, uint8_t *flags, size_t n, unsigned *ctr, KArgs<N> ka) {
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;
    constexpr int L = 8 * N;
    __attribute__((aligned(16))) uint8_t o[L];
    const uint8_t *a[KB], *z[KB];
#pragma unroll
    for (int j = 0; j < KB; j++) { a[j] = M.a[j] + ld * M.astride; z[j] = M.z[j] + ld * M.zstride; }
    const bool ok = GAL<N>::template gmulk_lane<KB>(o, a, z, zlen);
    if (idx < n) {
      flags[idx] = ok ? 0 : 1;
      if (ok) {
        uint4 *dst = reinterpret_cast<uint4 *>(out + idx * L);
        const uint4 *src = reinterpret_cast<const uint4 *>(o);
#pragma unroll
        for (int i = 0; i < L / 16; i++) dst[i] = src[i];
      }
    }
  }
}
// G1 of the 5-word fields (d159.param, f.param): the joint ladder in limb form (group_l5.cuh gmulk_lane)
template <class KP, int KB>
__global__ void __launch_bounds__(kBlock, 2) l5_gmulk_kernel(uint8_t *out, MultiArgs M, int zlen, uint8_t *flags, size_t n, KArgs<5> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = 2 * (size_t) fpk<5>().fbytes;
  const uint8_t *a[KB], *z[KB];
#pragma unroll
  for (int j = 0; j < KB; j++) { a[j] = M.a[j] + idx * M.astride; z[j] = M.z[j] + idx * M.zstride; }
  flags[idx] = GL<5, KP>::template gmulk_lane<KB>(out + idx * L, a, z, zlen) ? 0 : 1;
}
template <class G>
__global__ void __launch_bounds__(kBlock, 2) gt_multi_pow_kernel(uint8_t *out, MultiArgs M, int k, int zlen, size_t n, KArgs<G::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  gt_multi_pow_lane<G>(out + idx * (size_t) G::bytes(), M, idx, k, zlen);
}
// Z_r: the F_q micro-kernel of pbc_hip.hip on a constant block whose modulus is r (fp.cuh KArgs: only the FpK part is read).
// op 0 mul, 1 add, 2 sub, 3 invert, 4 neg, 5 halve, 6 double, 7 div (a / b), 8 element_from_hash (a: digests of `hlen` bytes)
template <int N>
__global__ void __launch_bounds__(kBlock) zr_op_kernel(int op, uint8_t *c, const uint8_t *a, const uint8_t *b, int hlen, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = fpk<N>().fbytes;
  zr_op_lane<N>(op, c + idx * L, a + idx * (op == 8 ? (size_t) hlen : L), b ? b + idx * L : nullptr, hlen);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static bool symmetric2(const pbc_hip_pairing_s *P) { return P->type == 'a' || P->type == '1' || P->type == 'e'; }
#define PBC_DISPATCH_G2(P_, group_, ...)                                                       \
  do {                                                                                         \
    if ((group_) == 2 && !symmetric2(P_)) PBC_DISPATCH_TWIST(P_, __VA_ARGS__);                 \
    else { PBC_DISPATCH_N((P_)->nlimb, { typedef FqOps<N> F; __VA_ARGS__; }); }                \
  } while (0)
#define PBC_DISPATCH_GT2(P_, ...)                                                              \
  do {                                                                                         \
    if ((P_)->type == 'a' || (P_)->type == '1') { if ((P_)->nlimb == 16) { typedef GtA<16> G; __VA_ARGS__; } else { typedef GtA<33> G; __VA_ARGS__; } } \
    else if ((P_)->type == 'e') { if ((P_)->nlimb == 16) { typedef GtE<16> G; __VA_ARGS__; } else { typedef GtE<33> G; __VA_ARGS__; } } \
    else if ((P_)->type == 'f') { PBC_DISPATCH_F((P_)->nlimb, { typedef GtF<N> G; __VA_ARGS__; }); } \
    else { PBC_DISPATCH_D(P_, { typedef GtD<N, DEG> G; __VA_ARGS__; }); }                      \
  } while (0)

static int check_obj(pbc_hip_pairing_s *P) {
  if (!P) return fail("null pairing");
  if (P->device < 0) return fail("no HIP device: libpbc_hip has no CPU fallback");
  return 0;
}
static int prepare(pbc_hip_pairing_s *P) {
  DeviceGuard guard(P->ndev > 0 ? P->devs[0] : P->device);
  return ensure_derived(P, 0);
}

// ---- the group law ------------------------------------------------------------------------------------------------------
static int affine_launch(pbc_hip_pairing_s *P, int op, int group, void *d_out, const void *d_a, const void *d_b, size_t n, hipStream_t s) {
  if (!n) return 0;
  const unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  PBC_DISPATCH_G2(P, group, hipLaunchKernelGGL(ec_affine_op_kernel<F>, dim3(grid), dim3(kBlock), 0, s, op, (uint8_t *) d_out, (const uint8_t *) d_a,
                                               (const uint8_t *) d_b, n, kargs<F::NW>(P)));
  HIP_TRY(hipGetLastError());
  return 0;
}
static int affine_host(pbc_hip_pairing_t *P, int op, int group, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n) {
  if (check_obj(P)) return 1;
  if (group != 1 && group != 2) return fail("group must be 1 or 2");
  if (!n) return 0;
  if (prepare(P)) return 1;
  const size_t lp = (size_t) (group == 2 ? P->len2 : P->len1);
  return run_host_generic(P, out, lp, a, lp, b, b ? lp : 0, n,
                          [P, op, group](void *d_out, const void *d_a, const void *d_b, size_t m, hipStream_t s, const OwnWs *) {
                            return affine_launch(P, op, group, d_out, d_a, d_b, m, s);
                          }, false);
}
static int affine_dev(pbc_hip_pairing_t *P, int op, int group, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream) {
  if (check_obj(P)) return 1;
  if (group != 1 && group != 2) return fail("group must be 1 or 2");
  if (ensure_derived(P, 0)) return 1;
  return affine_launch(P, op, group, d_out, d_a, d_b, n, (hipStream_t) stream);
}
extern "C" int pbc_hip_element_add_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n) {
  if (!a || !b) return fail("null argument");
  return affine_host(P, 0, group, out, a, b, n);
}
extern "C" int pbc_hip_element_sub_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n) {
  if (!a || !b) return fail("null argument");
  return affine_host(P, 1, group, out, a, b, n);
}
extern "C" int pbc_hip_element_neg_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *a, size_t n) {
  if (!a || !out) return fail("null argument");
  return affine_host(P, 2, group, out, a, nullptr, n);
}
extern "C" int pbc_hip_element_double_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *a, size_t n) {
  if (!a || !out) return fail("null argument");
  return affine_host(P, 3, group, out, a, nullptr, n);
}
extern "C" int pbc_hip_element_add_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream) {
  if (!d_a || !d_b) return fail("null argument");
  return affine_dev(P, 0, group, d_out, d_a, d_b, n, stream);
}
extern "C" int pbc_hip_element_sub_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream) {
  if (!d_a || !d_b) return fail("null argument");
  return affine_dev(P, 1, group, d_out, d_a, d_b, n, stream);
}
extern "C" int pbc_hip_element_neg_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_a, size_t n, void *stream) {
  if (!d_a || !d_out) return fail("null argument");
  return affine_dev(P, 2, group, d_out, d_a, nullptr, n, stream);
}
extern "C" int pbc_hip_element_double_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_a, size_t n, void *stream) {
  if (!d_a || !d_out) return fail("null argument");
  return affine_dev(P, 3, group, d_out, d_a, nullptr, n, stream);
}

// ---- multi-exponentiations ----------------------------------------------------------------------------------------------
static size_t group_len(const pbc_hip_pairing_s *P, int group) { return (size_t) (group == 1 ? P->len1 : group == 2 ? P->len2 : P->lenT); }
// Two routes.  DEFAULT: the composition of the library's tuned single-base kernels -- k x element_mul_zn / element_pow_zn
// (limb-form windowed ladders on G1, Lucas ladder / cyclotomic squarings on GT, ...) and k - 1 additions / products, all
// enqueued on the caller's stream with one stream-ordered temporary.  Measured on MI355X (profiles/r05_notes.md): the
// joint ladder over the generic word-form group law (below) does 5.4 M double scalar multiplications/s on a.param G1 and
// 19.6 M on d159 G1 where two single-base ladders + one addition do 10 M and 50 M; on GT the executed multiply-adds
// are 628 k against 2 x 196 k (a.param), 2.59 M against 2 x 760 k (f.param).  Shamir's trick saves doublings, but the
// single-base ladders save more by running in limb form / on the trace / in the cyclotomic subgroup.
// "hip_group_slow 1": the joint ladders -- ec_multi_mul_fast_kernel + ec_multi_mul_kernel for the lanes it reports, and
// gt_multi_pow_kernel -- which are also the independent route the tests compare the default with.
static int multi_launch(pbc_hip_pairing_s *P, int group, int k, void *d_out, const MultiArgs &M, size_t n, hipStream_t s, const OwnWs *own,
                        void *tmp_given = nullptr) {      // tmp_given: 2 n lp bytes the caller owns (host forms: no stream-ordered allocation)
  if (!n) return 0;
  const unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  uint8_t *o = (uint8_t *) d_out;
  const size_t lp = group_len(P, group);
  const bool joint_a = P->type == 'a' && !P->a_generic && group != 3;
  const bool joint_5 = group == 1 && P->nlimb == 5 && ((P->type == 'd' && P->deg == 3 && P->dconst.limb_ok) || (P->type == 'f' && P->fconst.pl_ok));
  if (!P->group_slow && !P->a_multi_compose && (joint_a || joint_5) && (k == 2 || k == 3)) {
    // type a on the 512-bit field, G1 of the five-word fields: ONE limb-form ladder for all bases (four doublings and k additions
    // per window; the composition doubles 4 k times).  A result that overlaps a base at another offset than its own is built in a
    // temporary: a lane reads its own records before it writes, other lanes' records must not change under them.
    bool overlap = false;
    for (int j = 0; j < k; j++) {
      const uint8_t *b = M.a[j];
      overlap |= b != o && b < o + n * lp && o < b + (n - 1) * M.astride + lp;
    }
    ProdWs W(P, s, own);
    uint8_t *flags = (uint8_t *) W.get(n + (overlap ? n * lp + 16 : 0));
    if (!flags) return 1;
    uint8_t *acc = overlap ? flags + ((n + 15) & ~(size_t) 15) : o;
    if (joint_a) {
      if (k == 2) hipLaunchKernelGGL((al_gmulk_kernel<16, 2>), dim3(PBC_RGRID(al_gmulk_kernel<16, 2>)), dim3(kBlock), 0, s, acc, M, P->len_zr, flags, n, unit_counter(P, s), kargs<16>(P));
      else hipLaunchKernelGGL((al_gmulk_kernel<16, 3>), dim3(PBC_RGRID(al_gmulk_kernel<16, 3>)), dim3(kBlock), 0, s, acc, M, P->len_zr, flags, n, unit_counter(P, s), kargs<16>(P));
      hipLaunchKernelGGL(ec_multi_mul_kernel<FqOps<16>>, dim3(grid), dim3(kBlock), 0, s, acc, M, k, P->len_zr, (const uint8_t *) flags, n, kargs<16>(P));
    } else {
      if (P->type == 'd' && k == 2) hipLaunchKernelGGL((l5_gmulk_kernel<KPd, 2>), dim3(grid), dim3(kBlock), 0, s, acc, M, P->len_zr, flags, n, kargs<5>(P));
      else if (P->type == 'd') hipLaunchKernelGGL((l5_gmulk_kernel<KPd, 3>), dim3(grid), dim3(kBlock), 0, s, acc, M, P->len_zr, flags, n, kargs<5>(P));
      else if (k == 2) hipLaunchKernelGGL((l5_gmulk_kernel<KPf, 2>), dim3(grid), dim3(kBlock), 0, s, acc, M, P->len_zr, flags, n, kargs<5>(P));
      else hipLaunchKernelGGL((l5_gmulk_kernel<KPf, 3>), dim3(grid), dim3(kBlock), 0, s, acc, M, P->len_zr, flags, n, kargs<5>(P));
      hipLaunchKernelGGL(ec_multi_mul_kernel<FqOps<5>>, dim3(grid), dim3(kBlock), 0, s, acc, M, k, P->len_zr, (const uint8_t *) flags, n, kargs<5>(P));
    }
    HIP_TRY(hipGetLastError());
    if (overlap) HIP_TRY(hipMemcpyAsync(d_out, acc, n * lp, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  if (!P->group_slow) {
    if (M.astride != lp || M.zstride != (size_t) P->len_zr) return fail("internal: packed records on the composition route");
    // `out` may be any ONE of the bases (include/pbc_hip.h; the reference lets x alias a base): the first term is written
    // before the later bases are read, so a result that overlaps a later base is accumulated in a second temporary
    bool alias = false;
    for (int j = 1; j < k; j++) {
      const uint8_t *b = M.a[j];
      alias |= b < o + n * lp && o < b + n * lp;
    }
    void *tmp = tmp_given, *acc = d_out;
    if (!tmp) HIP_TRY(hipMallocAsync(&tmp, n * lp * (alias ? 2 : 1), s));
    if (alias) acc = (uint8_t *) tmp + n * lp;
    int rc = 0;
    for (int j = 0; j < k && !rc; j++) {
      void *dst = j ? tmp : acc;
      rc = group == 3 ? pbc_hip_element_pow_zn_GT_batch_dev(P, dst, M.a[j], M.z[j], n, s)
                      : pbc_hip_element_mul_zn_batch_dev(P, group, dst, M.a[j], M.z[j], n, s);
      if (!rc && j)
        rc = group == 3 ? pbc_hip_element_mul_GT_batch_dev(P, acc, acc, tmp, n, s)
                        : pbc_hip_element_add_batch_dev(P, group, acc, acc, tmp, n, s);
    }
    if (!rc && alias && hipMemcpyAsync(d_out, acc, n * lp, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = fail("hipMemcpyAsync");
    if (!tmp_given) (void) hipFreeAsync(tmp, s);
    return rc;
  }
  if (group == 3) {
    PBC_DISPATCH_GT2(P, hipLaunchKernelGGL(gt_multi_pow_kernel<G>, dim3(grid), dim3(kBlock), 0, s, o, M, k, P->len_zr, n, kargs<G::NW>(P)));
  } else {
    ProdWs W(P, s, own);
    uint8_t *flags = (uint8_t *) W.get(n);
    if (!flags) return 1;
    PBC_DISPATCH_G2(P, group, {
      hipLaunchKernelGGL(ec_multi_mul_fast_kernel<F>, dim3(grid), dim3(kBlock), 0, s, o, M, k, P->len_zr, flags, n, kargs<F::NW>(P));
      hipLaunchKernelGGL(ec_multi_mul_kernel<F>, dim3(grid), dim3(kBlock), 0, s, o, M, k, P->len_zr, (const uint8_t *) flags, n, kargs<F::NW>(P));
    });
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
// host buffers: the k bases and the k scalars of a unit are packed side by side (k lp and k lz bytes per unit), so that the
// host-buffer path of the pairings (device set, chunk ring) serves them as one two-input operation
static int multi_host(pbc_hip_pairing_t *P, int group, int k, uint8_t *out, const uint8_t *const *a, const uint8_t *const *z, size_t n) {
  if (check_obj(P)) return 1;
  if (group < 1 || group > 3) return fail("group must be 1, 2 or 3 (GT)");
  for (int j = 0; j < k; j++)
    if (!a[j] || !z[j]) return fail("null argument");
  if (!n) return 0;
  if (prepare(P)) return 1;
  const size_t lp = group_len(P, group), lz = (size_t) P->len_zr;
  if (!P->group_slow) {
    // the composition of the single-base kernels wants each base / scalar array by itself: staged on the object's first
    // device
    DeviceGuard guard(P->ndev > 0 ? P->devs[0] : P->device);
    DevBuf ba[3], bz[3], bo;
    const void *da[3] = {nullptr, nullptr, nullptr}, *dz[3] = {nullptr, nullptr, nullptr};
    for (int j = 0; j < k; j++) {
      HIP_TRY(ba[j].alloc(n * lp));
      HIP_TRY(bz[j].alloc(n * lz));
      HIP_TRY(hipMemcpy(ba[j].p, a[j], n * lp, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(bz[j].p, z[j], n * lz, hipMemcpyHostToDevice));
      da[j] = ba[j].p;
      dz[j] = bz[j].p;
    }
    HIP_TRY(bo.alloc(n * lp));
    DevBuf btmp;                        // (the temporaries of the composition: owned here -- the stream-ordered allocator is kept
    HIP_TRY(btmp.alloc(2 * n * lp));    //  off the host forms, pbc_hip.hip "ScratchEnt")
    MultiArgs M;
    for (int j = 0; j < 3; j++) {
      M.a[j] = (const uint8_t *) da[j < k ? j : 0];
      M.z[j] = (const uint8_t *) dz[j < k ? j : 0];
    }
    M.astride = lp;
    M.zstride = lz;
    if (multi_launch(P, group, k, bo.p, M, n, 0, nullptr, btmp.p)) return 1;
    HIP_TRY(hipStreamSynchronize(0));
    HIP_TRY(hipMemcpy(out, bo.p, n * lp, hipMemcpyDeviceToHost));
    return 0;
  }
  std::vector<uint8_t> A(n * lp * k), Z(n * lz * k);
  for (size_t i = 0; i < n; i++)
    for (int j = 0; j < k; j++) {
      memcpy(&A[(i * k + j) * lp], a[j] + i * lp, lp);
      memcpy(&Z[(i * k + j) * lz], z[j] + i * lz, lz);
    }
  return run_host_generic(P, out, lp, A.data(), lp * k, Z.data(), lz * k, n,
                          [P, group, k, lp, lz](void *d_out, const void *d_a, const void *d_b, size_t m, hipStream_t s, const OwnWs *own) {
                            MultiArgs M;
                            for (int j = 0; j < 3; j++) {
                              M.a[j] = (const uint8_t *) d_a + (j < k ? j : 0) * lp;
                              M.z[j] = (const uint8_t *) d_b + (j < k ? j : 0) * lz;
                            }
                            M.astride = lp * k;
                            M.zstride = lz * k;
                            return multi_launch(P, group, k, d_out, M, m, s, own);
                          }, false);
}
static int multi_dev(pbc_hip_pairing_t *P, int group, int k, void *d_out, const void *const *a, const void *const *z, size_t n, void *stream) {
  if (check_obj(P)) return 1;
  if (group < 1 || group > 3) return fail("group must be 1, 2 or 3 (GT)");
  for (int j = 0; j < k; j++)
    if (!a[j] || !z[j]) return fail("null argument");
  if (ensure_derived(P, 0)) return 1;
  MultiArgs M;
  for (int j = 0; j < 3; j++) {
    M.a[j] = (const uint8_t *) a[j < k ? j : 0];
    M.z[j] = (const uint8_t *) z[j < k ? j : 0];
  }
  M.astride = group_len(P, group);
  M.zstride = (size_t) P->len_zr;
  return multi_launch(P, group, k, d_out, M, n, (hipStream_t) stream, nullptr);
}
extern "C" int pbc_hip_element_pow2_zn_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *a1, const uint8_t *n1,
                                             const uint8_t *a2, const uint8_t *n2, size_t n) {
  const uint8_t *a[3] = {a1, a2, nullptr}, *z[3] = {n1, n2, nullptr};
  return multi_host(P, group, 2, out, a, z, n);
}
extern "C" int pbc_hip_element_pow3_zn_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *a1, const uint8_t *n1,
                                             const uint8_t *a2, const uint8_t *n2, const uint8_t *a3, const uint8_t *n3, size_t n) {
  const uint8_t *a[3] = {a1, a2, a3}, *z[3] = {n1, n2, n3};
  return multi_host(P, group, 3, out, a, z, n);
}
extern "C" int pbc_hip_element_pow2_zn_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_a1, const void *d_n1,
                                                 const void *d_a2, const void *d_n2, size_t n, void *stream) {
  const void *a[3] = {d_a1, d_a2, nullptr}, *z[3] = {d_n1, d_n2, nullptr};
  return multi_dev(P, group, 2, d_out, a, z, n, stream);
}
extern "C" int pbc_hip_element_pow3_zn_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_a1, const void *d_n1,
                                                 const void *d_a2, const void *d_n2, const void *d_a3, const void *d_n3, size_t n, void *stream) {
  const void *a[3] = {d_a1, d_a2, d_a3}, *z[3] = {d_n1, d_n2, d_n3};
  return multi_dev(P, group, 3, d_out, a, z, n, stream);
}

// ---- Z_r ------------------------------------------------------------------------------------------------------------------
static int zr_launch(pbc_hip_pairing_s *P, int op, int hlen, void *d_out, const void *d_a, const void *d_b, size_t n, hipStream_t s) {
  if (!n) return 0;
  const unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  PBC_DISPATCH_N(P->zr_nlimb, {
    KArgs<N> K;
    if (zr_kargs<N>(P, K)) return 1;
    hipLaunchKernelGGL(zr_op_kernel<N>, dim3(grid), dim3(kBlock), 0, s, op, (uint8_t *) d_out, (const uint8_t *) d_a, (const uint8_t *) d_b, hlen, n, K);
  });
  HIP_TRY(hipGetLastError());
  return 0;
}
static int zr_check(pbc_hip_pairing_s *P, int op, const void *a, const void *b, int hlen) {
  if (check_obj(P)) return 1;
  if (op < 0 || op > 8) return fail("Z_r: bad op");
  if (!a) return fail("null argument");
  const bool binary = op <= 2 || op == 7;
  if (binary && !b) return fail("Z_r: this operation takes two operands");
  if (op == 8 && hlen < 1) return fail("hlen must be >= 1");
  if (!P->zr_nlimb) return fail("Z_r: no arithmetic for this group order");
  return 0;
}
extern "C" int pbc_hip_zr_op_batch(pbc_hip_pairing_t *P, int op, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n) {
  if (zr_check(P, op, a, b, 1)) return 1;
  if (op == 8) return fail("Z_r: element_from_hash has its own entry point");
  if (!n) return 0;
  const bool binary = op <= 2 || op == 7;
  const size_t lz = (size_t) P->len_zr;
  return run_host_generic(P, out, lz, a, lz, binary ? b : nullptr, binary ? lz : 0, n,
                          [P, op](void *d_out, const void *d_a, const void *d_b, size_t m, hipStream_t s, const OwnWs *) {
                            return zr_launch(P, op, 0, d_out, d_a, d_b, m, s);
                          }, false);
}
extern "C" int pbc_hip_zr_op_batch_dev(pbc_hip_pairing_t *P, int op, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream) {
  if (zr_check(P, op, d_a, d_b, 1)) return 1;
  if (op == 8) return fail("Z_r: element_from_hash has its own entry point");
  const bool binary = op <= 2 || op == 7;
  return zr_launch(P, op, 0, d_out, d_a, binary ? d_b : nullptr, n, (hipStream_t) stream);
}
extern "C" int pbc_hip_zr_from_hash_batch(pbc_hip_pairing_t *P, uint8_t *out, const uint8_t *data, int hlen, size_t n) {
  if (zr_check(P, 8, data, nullptr, hlen)) return 1;
  if (!n) return 0;
  return run_host_generic(P, out, (size_t) P->len_zr, data, (size_t) hlen, nullptr, 0, n,
                          [P, hlen](void *d_out, const void *d_a, const void *, size_t m, hipStream_t s, const OwnWs *) {
                            return zr_launch(P, 8, hlen, d_out, d_a, nullptr, m, s);
                          }, false);
}
extern "C" int pbc_hip_zr_from_hash_batch_dev(pbc_hip_pairing_t *P, void *d_out, const void *d_data, int hlen, size_t n, void *stream) {
  if (zr_check(P, 8, d_data, nullptr, hlen)) return 1;
  return zr_launch(P, 8, hlen, d_out, d_data, nullptr, n, (hipStream_t) stream);
}
