// pbc_hip_group.hip -- kernels and C-ABI entry points of the group operations next to the pairing: libpbc_hip.so; see host_common.h
#include "host_common.h"

// ---- group operations (one element per lane) -------------------------------------------------
// element_mul_zn: the COMPLETE ladder (ec_mul_lane: any point, any scalar, two group operations per bit) over a field
// policy F -- FqOps for G1 and the G2 of symmetric types, FdOps / Fq2Ops for the twists.  flags != null: only the lanes a
// fast kernel reported (flags[i] != 0) run; a wave without such a lane retires at once.  in_stride 0: every lane takes the
// same point (the complete pass behind a fixed-base table).
template <class F>
__global__ void __launch_bounds__(kBlock, 2) ec_mul_kernel(uint8_t *out, const uint8_t *in, size_t in_stride, const uint8_t *z,
                                                            int zlen, const uint8_t *flags, size_t n, KArgs<F::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  if (flags && !flags[idx]) return;
  const size_t L = 2 * (size_t) F::bytes();
  ec_mul_lane<F>(out + idx * L, in + idx * in_stride, z + idx * zlen, zlen);
}
// The regular signed-window ladder over a field policy (ec_mul_win_lane); lanes it cannot finish are flagged.
template <class F>
__global__ void __launch_bounds__(kBlock, 2) ec_mul_win_kernel(uint8_t *out, const uint8_t *in, const uint8_t *z, int zlen,
                                                                uint8_t *flags, size_t n, KArgs<F::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = 2 * (size_t) F::bytes();
  flags[idx] = ec_mul_win_lane<F>(out + idx * L, in + idx * L, z + idx * zlen, zlen) ? 0 : 1;
}
// G1 of the 5-word fields (d159.param, f.param): the same ladder in limb form (group_l5.cuh)
template <class KP>
__global__ void __launch_bounds__(kBlock, 2) l5_gmul_kernel(uint8_t *out, const uint8_t *in, const uint8_t *z, int zlen, uint8_t *flags, size_t n, KArgs<5> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = 2 * (size_t) fpk<5>().fbytes;
  flags[idx] = GL<5, KP>::gmul_lane(out + idx * L, in + idx * L, z + idx * zlen, zlen) ? 0 : 1;
}
template <class KP>
__global__ void __launch_bounds__(kBlock, 2) l5_pp_pow_kernel(uint8_t *out, const uint32_t *__restrict__ tab, const uint8_t *z, int zlen, uint8_t *flags, size_t n, KArgs<5> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = 2 * (size_t) fpk<5>().fbytes;
  flags[idx] = GL<5, KP>::pp_pow_lane(out + idx * L, tab, z + idx * zlen, zlen) ? 0 : 1;
}
// Type a, 512-bit field: the same ladder on the limb-form arithmetic (group_al.cuh), resident workgroups as the pairing kernel.
template <int N>
__global__ void __launch_bounds__(kBlock, PBC_A_WAVES) al_gmul_kernel(uint8_t *out, const uint8_t *in, const uint8_t *z, int zlen,
                                                                       uint8_t *flags, size_t n, unsigned *ctr, KArgs<N> ka) {
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;
    constexpr int L = 8 * N;
    __attribute__((aligned(16))) uint8_t o[L];
    const bool ok = GAL<N>::gmul_lane(o, in + ld * L, z + ld * zlen, zlen);
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
// element_pow_zn on GT, type a: the Lucas ladder for elements of norm 1 (group_al.cuh); others are flagged for gt_op_kernel
template <int N>
__global__ void __launch_bounds__(kBlock, PBC_A_WAVES) al_gtpow_kernel(uint8_t *out, const uint8_t *a, const uint8_t *z, int zlen,
                                                                        uint8_t *flags, size_t n, unsigned *ctr, KArgs<N> ka) {
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;
    constexpr int L = 8 * N;
    __attribute__((aligned(16))) uint8_t o[L];
    const bool ok = GAL<N>::gt_pow_lane(o, a + ld * L, z + ld * zlen, zlen);
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
// element_from_hash, type a: the limb-form routine (group_al.cuh from_hash_lane); flagged lanes: g_from_hash_kernel
template <int N>
__global__ void __launch_bounds__(kBlock, PBC_A_WAVES) al_hash_kernel(uint8_t *out, const uint8_t *data, int hlen, uint8_t *flags, size_t n,
                                                                       unsigned *ctr, KArgs<N> ka) {
  PBC_RESIDENT_LOOP(n, ctr) {
    size_t idx = PBC_UNIT_INDEX;
    size_t ld = idx < n ? idx : n - 1;   // whole waves search together
    constexpr int L = 8 * N;
    __attribute__((aligned(16))) uint8_t o[L];
    const bool ok = GAL<N>::from_hash_lane(o, data + ld * hlen, hlen);
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
// element_pow_zn on GT, type f (5-word fields): cyclotomic squarings in the LDS area (group_ops.cuh f_gt_pow_cyc_lane);
// elements outside the cyclotomic subgroup are flagged for gt_op_kernel
template <int N, bool BM1, bool XS>
__global__ void __launch_bounds__(kBlock, 2) f_gtpow_kernel(uint8_t *out, const uint8_t *a, const uint8_t *z, int zlen, uint8_t *flags, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  size_t ld = idx < n ? idx : n - 1;
  const int LT = 12 * (int) fpk<N>().fbytes;
  __attribute__((aligned(4))) uint8_t o[48 * N];
  const bool ok = f_gt_pow_cyc_lane<TypeF<N, BM1, XS>>(o, a + ld * LT, z + ld * zlen, zlen);
  if (idx < n) {
    flags[idx] = ok ? 0 : 1;
    if (ok) {
      uint32_t *dst = reinterpret_cast<uint32_t *>(out + idx * LT);
      const uint32_t *src = reinterpret_cast<const uint32_t *>(o);
      for (int i = 0; i < LT / 4; i++) dst[i] = src[i];
    }
  }
}
// fixed-base tables (group_ops.cuh ec_pp_* / gt_pp_*): one entry per lane; one power per lane
template <class F>
__global__ void __launch_bounds__(kBlock, 2) ec_pp_init_kernel(uint32_t *tab, uint8_t *flags, const uint8_t *base, int zlen, size_t units, KArgs<F::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= units) return;
  ec_pp_entry_lane<F>(tab, flags, base, zlen, idx);
}
template <class F>
__global__ void __launch_bounds__(kBlock, 2) ec_pp_pow_kernel(uint8_t *out, const uint32_t *__restrict__ tab, const uint8_t *z, int zlen,
                                                               uint8_t *flags, size_t n, KArgs<F::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  const size_t L = 2 * (size_t) F::bytes();
  flags[idx] = ec_pp_pow_lane<F>(out + idx * L, tab, z + idx * zlen, zlen) ? 0 : 1;
}
template <class G>
__global__ void __launch_bounds__(kBlock, 2) gt_pp_init_kernel(uint32_t *tab, const uint8_t *a, size_t units, KArgs<G::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= units) return;
  gt_pp_entry_lane<G>(tab, a, idx);
}
template <class G>
__global__ void __launch_bounds__(kBlock, 2) gt_pp_pow_kernel(uint8_t *out, const uint32_t *__restrict__ tab, const uint8_t *z, int zlen, size_t n, KArgs<G::NW> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  gt_pp_pow_lane<G>(out + idx * (size_t) G::bytes(), tab, z + idx * zlen, zlen);
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
template <int N>
__global__ void __launch_bounds__(kBlock, 2) g_from_hash_kernel(uint8_t *out, const uint8_t *data, int hlen, const uint8_t *flags, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  size_t ld = idx < n ? idx : n - 1;   // whole waves search together
  // (flags != null: the complete pass behind al_hash_kernel -- a wave without a reported lane retires at once, else the
  // whole wave searches, since the search deals its candidates over all 64 lanes, and the reported lanes store)
  const bool mine = !flags || (idx < n && flags[idx] != 0);
  if (flags && !__ballot(mine)) return;
  const int L = 2 * (int) fpk<N>().fbytes;
  __attribute__((aligned(4))) uint8_t o[8 * N];
  g_from_hash_lane<N>(o, data + ld * hlen, hlen);
  if (idx < n && mine)
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
                                                           const uint8_t *b, int lenT, int zlen, const uint8_t *flags, size_t n, KArgs<N> ka) {
  size_t idx = (size_t) blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n) return;
  if (flags && !flags[idx]) return;    // the complete pass behind al_gtpow_kernel
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


// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
extern "C" int pbc_hip_pairing_length_in_bytes_Zr(const pbc_hip_pairing_t *p) { return p->len_zr; }
extern "C" int pbc_hip_pairing_length_in_bytes_compressed_G1(const pbc_hip_pairing_t *p) { return p->len_fq + 1; }
extern "C" int pbc_hip_pairing_length_in_bytes_compressed_G2(const pbc_hip_pairing_t *p) { return p->len2 / 2 + 1; }
extern "C" int pbc_hip_pairing_length_in_bytes_x_only_G1(const pbc_hip_pairing_t *p) { return p->len_fq; }
extern "C" int pbc_hip_pairing_length_in_bytes_x_only_G2(const pbc_hip_pairing_t *p) { return p->len2 / 2; }

static bool symmetric(const pbc_hip_pairing_s *P) { return P->type == 'a' || P->type == '1' || P->type == 'e'; }
// F = the field policy of group 1 / 2 of this pairing: F_q for G1 and the G2 of the symmetric types, the twist's field else
#define PBC_DISPATCH_G(P_, group_, ...)                                                        \
  do {                                                                                         \
    if ((group_) == 2 && !symmetric(P_)) PBC_DISPATCH_TWIST(P_, __VA_ARGS__);                  \
    else { PBC_DISPATCH_N((P_)->nlimb, { typedef FqOps<N> F; __VA_ARGS__; }); }                \
  } while (0)
// G = GT as a field policy
#define PBC_DISPATCH_GT(P_, ...)                                                               \
  do {                                                                                         \
    if ((P_)->type == 'a' || (P_)->type == '1') { if ((P_)->nlimb == 16) { typedef GtA<16> G; __VA_ARGS__; } else { typedef GtA<33> G; __VA_ARGS__; } } \
    else if ((P_)->type == 'e') { if ((P_)->nlimb == 16) { typedef GtE<16> G; __VA_ARGS__; } else { typedef GtE<33> G; __VA_ARGS__; } } \
    else if ((P_)->type == 'f') { PBC_DISPATCH_F((P_)->nlimb, { typedef GtF<N> G; __VA_ARGS__; }); } \
    else { PBC_DISPATCH_D(P_, { typedef GtD<N, DEG> G; __VA_ARGS__; }); }                      \
  } while (0)

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

// The operations of this file behind one launcher.  Records: a (la bytes per unit), b (lb; 0: none), out (lo).
enum GroupOp { G_MUL, GT_MUL, GT_POW, GT_FINALPOW, G_HASH, G_COMPRESS, G_DECOMPRESS, G_TO_X, G_FROM_X };
struct GroupCall {
  int op, group, aux;                  // aux: digest length of G_HASH
  size_t la, lb, lo;
};
static int group_call(const pbc_hip_pairing_s *P, GroupCall &c, int op, int group, int aux) {
  c.op = op; c.group = group; c.aux = aux;
  const size_t lp = (size_t) (group == 2 ? P->len2 : P->len1), lc = lp / 2 + 1, lx = lp / 2, lt = (size_t) P->lenT, lz = (size_t) P->len_zr;
  if (op == G_MUL || op >= G_HASH)
    if (group != 1 && group != 2) return fail("group must be 1 or 2");
  switch (op) {
    case G_MUL: c.la = c.lo = lp; c.lb = lz; break;
    case GT_MUL: c.la = c.lb = c.lo = lt; break;
    case GT_POW: c.la = c.lo = lt; c.lb = lz; break;
    case GT_FINALPOW: c.la = c.lo = lt; c.lb = 0; break;
    case G_HASH:
      if (aux < 1) return fail("hlen must be >= 1");
      if (group == 2 && P->type == 'f' && aux < 2) return fail("type f G2: hlen must be >= 2 (fq_from_hash halves the digest)");
      c.la = (size_t) aux; c.lb = 0; c.lo = lp; break;
    case G_COMPRESS: c.la = lp; c.lb = 0; c.lo = lc; break;
    case G_DECOMPRESS: c.la = lc; c.lb = 0; c.lo = lp; break;
    case G_TO_X: c.la = lp; c.lb = 0; c.lo = lx; break;
    case G_FROM_X: c.la = lx; c.lb = 0; c.lo = lp; break;
    default: return fail("internal: unknown group operation");
  }
  return 0;
}
// one-time constants an operation needs (single-lane kernels on the default stream of the current device; first use only)
static int group_prepare(pbc_hip_pairing_s *P, const GroupCall &c) {
  if (P->device < 0) return fail("no HIP device: libpbc_hip has no CPU fallback");
  if (ensure_derived(P, 0)) return 1;
  if (c.op >= G_HASH) {
    const bool twist = c.group == 2 && !symmetric(P);
    if (twist && ensure_ext_sqrt(P)) return 1;
    if ((!twist || P->type == 'f') && ensure_sqrt_constants(P)) return 1;   // type f: fq_sqrt works through square roots in F_q
  }
  return 0;
}
// enqueue n units on stream s (device pointers); `own`: the workspace of a host-path stream, else the object's table
static int group_launch(pbc_hip_pairing_s *P, const GroupCall &c, void *d_out, const void *d_a, const void *d_b, size_t n,
                        hipStream_t s, const OwnWs *own) {
  if (!n) return 0;
  uint8_t *o = (uint8_t *) d_out;
  const uint8_t *a = (const uint8_t *) d_a, *b = (const uint8_t *) d_b;
  const unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  ProdWs W(P, s, own);
  const bool fast_a = P->type == 'a' && !P->a_generic && !P->group_slow;
  if (c.op == G_MUL) {
    if (P->group_slow) {
      PBC_DISPATCH_G(P, c.group, hipLaunchKernelGGL(ec_mul_kernel<F>, dim3(grid), dim3(kBlock), 0, s, o, a, c.la, b, P->len_zr, (const uint8_t *) nullptr, n, kargs<F::NW>(P)));
    } else {
      uint8_t *flags = (uint8_t *) W.get(n);
      if (!flags) return 1;
      if (fast_a) {
        hipLaunchKernelGGL(al_gmul_kernel<16>, dim3(PBC_RGRID(al_gmul_kernel<16>)), dim3(kBlock), 0, s, o, a, b, P->len_zr, flags, n, unit_counter(P, s), kargs<16>(P));
        hipLaunchKernelGGL(ec_mul_kernel<FqOps<16>>, dim3(grid), dim3(kBlock), 0, s, o, a, c.la, b, P->len_zr, (const uint8_t *) flags, n, kargs<16>(P));
      } else if (c.group == 1 && P->nlimb == 5 && ((P->type == 'd' && P->deg == 3 && P->dconst.limb_ok) || (P->type == 'f' && P->fconst.pl_ok))) {
        // (the borrowed constants of the pairing kernels' limb-form steps fit this q: host_params.h limb_ok / pl_ok)
        if (P->type == 'd') hipLaunchKernelGGL(l5_gmul_kernel<KPd>, dim3(grid), dim3(kBlock), 0, s, o, a, b, P->len_zr, flags, n, kargs<5>(P));
        else hipLaunchKernelGGL(l5_gmul_kernel<KPf>, dim3(grid), dim3(kBlock), 0, s, o, a, b, P->len_zr, flags, n, kargs<5>(P));
        hipLaunchKernelGGL(ec_mul_kernel<FqOps<5>>, dim3(grid), dim3(kBlock), 0, s, o, a, c.la, b, P->len_zr, (const uint8_t *) flags, n, kargs<5>(P));
      } else {
        PBC_DISPATCH_G(P, c.group, {
          hipLaunchKernelGGL(ec_mul_win_kernel<F>, dim3(grid), dim3(kBlock), 0, s, o, a, b, P->len_zr, flags, n, kargs<F::NW>(P));
          hipLaunchKernelGGL(ec_mul_kernel<F>, dim3(grid), dim3(kBlock), 0, s, o, a, c.la, b, P->len_zr, (const uint8_t *) flags, n, kargs<F::NW>(P));
        });
      }
    }
  } else if (c.op == GT_POW && fast_a) {
    uint8_t *flags = (uint8_t *) W.get(n);
    if (!flags) return 1;
    hipLaunchKernelGGL(al_gtpow_kernel<16>, dim3(PBC_RGRID(al_gtpow_kernel<16>)), dim3(kBlock), 0, s, o, a, b, P->len_zr, flags, n, unit_counter(P, s), kargs<16>(P));
    hipLaunchKernelGGL(gt_op_kernel<16>, dim3(grid), dim3(kBlock), 0, s, P->type, 1, o, a, b, P->lenT, P->len_zr, (const uint8_t *) flags, n, kargs<16>(P));
  } else if (c.op == GT_POW && P->type == 'f' && P->nlimb == 5 && !P->group_slow) {
    uint8_t *flags = (uint8_t *) W.get(n);
    if (!flags) return 1;
    if (P->f_bm1 && P->fconst_i.xs_ok)
      hipLaunchKernelGGL((f_gtpow_kernel<5, true, true>), dim3(grid), dim3(kBlock), 0, s, o, a, b, P->len_zr, flags, n, kargs<5>(P, true));
    else if (P->f_bm1)
      hipLaunchKernelGGL((f_gtpow_kernel<5, true, false>), dim3(grid), dim3(kBlock), 0, s, o, a, b, P->len_zr, flags, n, kargs<5>(P, true));
    else
      hipLaunchKernelGGL((f_gtpow_kernel<5, false, false>), dim3(grid), dim3(kBlock), 0, s, o, a, b, P->len_zr, flags, n, kargs<5>(P));
    hipLaunchKernelGGL(gt_op_kernel<5>, dim3(grid), dim3(kBlock), 0, s, P->type, 1, o, a, b, P->lenT, P->len_zr, (const uint8_t *) flags, n, kargs<5>(P));
  } else if (c.op == GT_MUL || c.op == GT_POW || c.op == GT_FINALPOW) {
    PBC_DISPATCH_N(P->nlimb, hipLaunchKernelGGL(gt_op_kernel<N>, dim3(grid), dim3(kBlock), 0, s, P->type, c.op - GT_MUL, o, a, b, P->lenT,
                                                P->len_zr, (const uint8_t *) nullptr, n, kargs<N>(P)));
  } else if (c.group == 2 && !symmetric(P)) {          // point formats and hashing on the twists
    const int what = c.op == G_HASH ? 0 : c.op - G_COMPRESS + 1;
    PBC_DISPATCH_TWIST(P, hipLaunchKernelGGL(g2_point_kernel<F>, dim3(grid), dim3(kBlock), 0, s, what, o, a, c.aux, n, kargs<F::NW>(P)));
  } else if (c.op == G_HASH && fast_a) {
    uint8_t *flags = (uint8_t *) W.get(n);
    if (!flags) return 1;
    hipLaunchKernelGGL(al_hash_kernel<16>, dim3(PBC_RGRID(al_hash_kernel<16>)), dim3(kBlock), 0, s, o, a, c.aux, flags, n, unit_counter(P, s), kargs<16>(P));
    hipLaunchKernelGGL(g_from_hash_kernel<16>, dim3(grid), dim3(kBlock), 0, s, o, a, c.aux, (const uint8_t *) flags, n, kargs<16>(P));
  } else if (c.op == G_HASH) {
    PBC_DISPATCH_N(P->nlimb, hipLaunchKernelGGL(g_from_hash_kernel<N>, dim3(grid), dim3(kBlock), 0, s, o, a, c.aux, (const uint8_t *) nullptr, n, kargs<N>(P)));
  } else {
    PBC_DISPATCH_N(P->nlimb, hipLaunchKernelGGL(g_compress_kernel<N>, dim3(grid), dim3(kBlock), 0, s, c.op - G_COMPRESS, o, a, n, kargs<N>(P)));
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
// host buffers: the host-buffer path of the pairings (device set, page-locked buffers in place, else staged chunk buffers
// the object keeps).  In place only where the kernels read their records once and with word loads.
static int group_host(pbc_hip_pairing_t *P, int op, int group, int aux, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n) {
  if (!P) return fail("null pairing");
  GroupCall c;
  if (group_call(P, c, op, group, aux)) return 1;
  if (!n) return P->device < 0 ? fail("no HIP device: libpbc_hip has no CPU fallback") : 0;
  {
    DeviceGuard guard(P->ndev > 0 ? P->devs[0] : P->device);
    if (group_prepare(P, c)) return 1;
  }
  // in place over PCIe only where a lane reads its records ONCE and with word loads: GT products and final powers, and the
  // limb-form type a ladders (scalars of a whole number of words); the bit-by-bit routines read a scalar byte per step
  const bool fast_a = P->type == 'a' && !P->a_generic && !P->group_slow && P->len_zr % 4 == 0;
  const bool in_place = op == GT_MUL || op == GT_FINALPOW || ((op == G_MUL || op == GT_POW) && fast_a);
  return run_host_generic(P, out, c.lo, a, c.la, c.lb ? b : nullptr, c.lb, n,
                          [P, c](void *d_out, const void *d_a, const void *d_b, size_t m, hipStream_t s, const OwnWs *own) {
                            return group_launch(P, c, d_out, d_a, d_b, m, s, own);
                          }, in_place);
}
static int group_dev(pbc_hip_pairing_t *P, int op, int group, int aux, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream) {
  if (!P) return fail("null pairing");
  GroupCall c;
  if (group_call(P, c, op, group, aux)) return 1;
  if (group_prepare(P, c)) return 1;
  return group_launch(P, c, d_out, d_a, d_b, n, (hipStream_t) stream, nullptr);
}

extern "C" int pbc_hip_element_mul_zn_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *in, const uint8_t *zr, size_t n) {
  return group_host(P, G_MUL, group, 0, out, in, zr, n);
}
extern "C" int pbc_hip_element_mul_zn_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_in, const void *d_zr, size_t n, void *stream) {
  return group_dev(P, G_MUL, group, 0, d_out, d_in, d_zr, n, stream);
}
extern "C" int pbc_hip_element_mul_GT_batch(pbc_hip_pairing_t *P, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n) {
  return group_host(P, GT_MUL, 0, 0, out, a, b, n);
}
extern "C" int pbc_hip_element_mul_GT_batch_dev(pbc_hip_pairing_t *P, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream) {
  return group_dev(P, GT_MUL, 0, 0, d_out, d_a, d_b, n, stream);
}
extern "C" int pbc_hip_element_pow_zn_GT_batch(pbc_hip_pairing_t *P, uint8_t *out, const uint8_t *a, const uint8_t *zr, size_t n) {
  return group_host(P, GT_POW, 0, 0, out, a, zr, n);
}
extern "C" int pbc_hip_element_pow_zn_GT_batch_dev(pbc_hip_pairing_t *P, void *d_out, const void *d_a, const void *d_zr, size_t n, void *stream) {
  return group_dev(P, GT_POW, 0, 0, d_out, d_a, d_zr, n, stream);
}
extern "C" int pbc_hip_finalpow_batch(pbc_hip_pairing_t *P, uint8_t *out, const uint8_t *in, size_t n) {
  return group_host(P, GT_FINALPOW, 0, 0, out, in, nullptr, n);
}
extern "C" int pbc_hip_finalpow_batch_dev(pbc_hip_pairing_t *P, void *d_out, const void *d_in, size_t n, void *stream) {
  return group_dev(P, GT_FINALPOW, 0, 0, d_out, d_in, nullptr, n, stream);
}
extern "C" int pbc_hip_element_from_hash_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *data, int hlen, size_t n) {
  return group_host(P, G_HASH, group, hlen, out, data, nullptr, n);
}
extern "C" int pbc_hip_element_from_hash_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_data, int hlen, size_t n, void *stream) {
  return group_dev(P, G_HASH, group, hlen, d_out, d_data, nullptr, n, stream);
}
extern "C" int pbc_hip_element_to_bytes_compressed_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *in, size_t n) {
  return group_host(P, G_COMPRESS, group, 0, out, in, nullptr, n);
}
extern "C" int pbc_hip_element_to_bytes_compressed_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_in, size_t n, void *stream) {
  return group_dev(P, G_COMPRESS, group, 0, d_out, d_in, nullptr, n, stream);
}
extern "C" int pbc_hip_element_from_bytes_compressed_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *in, size_t n) {
  return group_host(P, G_DECOMPRESS, group, 0, out, in, nullptr, n);
}
extern "C" int pbc_hip_element_from_bytes_compressed_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_in, size_t n, void *stream) {
  return group_dev(P, G_DECOMPRESS, group, 0, d_out, d_in, nullptr, n, stream);
}
extern "C" int pbc_hip_element_to_bytes_x_only_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *in, size_t n) {
  return group_host(P, G_TO_X, group, 0, out, in, nullptr, n);
}
extern "C" int pbc_hip_element_to_bytes_x_only_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_in, size_t n, void *stream) {
  return group_dev(P, G_TO_X, group, 0, d_out, d_in, nullptr, n, stream);
}
extern "C" int pbc_hip_element_from_bytes_x_only_batch(pbc_hip_pairing_t *P, int group, uint8_t *out, const uint8_t *in, size_t n) {
  return group_host(P, G_FROM_X, group, 0, out, in, nullptr, n);
}
extern "C" int pbc_hip_element_from_bytes_x_only_batch_dev(pbc_hip_pairing_t *P, int group, void *d_out, const void *d_in, size_t n, void *stream) {
  return group_dev(P, G_FROM_X, group, 0, d_out, d_in, nullptr, n, stream);
}

// ---- fixed-base powers: element_pp_init / element_pp_pow_zn / element_pp_clear (include/pbc_field.h:591-625) -----------
struct pbc_hip_element_pp_s {
  pbc_hip_pairing_s *P;
  int group;                           // 1, 2: scalar multiples of a point; 3: powers of a GT element
  int device;
  uint32_t *tab;                       // device: group 1 / 2 [len_zr][255][2] elements, group 3 [len_zr][256] elements (Montgomery words)
  uint8_t *base;                       // device: the base's record (the complete pass of group 1 / 2 reads it)
  bool complete_only;                  // group 1 / 2: the base has small order (a table entry is O) or is off the curve: every power takes the complete ladder
};
extern "C" int pbc_hip_element_pp_init(pbc_hip_element_pp_t **out, pbc_hip_pairing_t *P, int group, const uint8_t *in) {
  if (!out || !P || !in) return fail("null argument");
  if (group < 1 || group > 3) return fail("element_pp_init: group must be 1, 2 or 3 (GT)");
  if (P->device < 0) return fail("no HIP device: libpbc_hip has no CPU fallback");
  const int dev = P->device;           // the table lives on the device the pairing object was created on, whatever is current
  DeviceGuard guard(dev);              // (as pbc_hip_pairing_pp_init)
  if (ensure_derived(P, 0)) return 1;
  const size_t lrec = group == 3 ? (size_t) P->lenT : group == 2 ? (size_t) P->len2 : (size_t) P->len1;
  const size_t rows = (size_t) P->len_zr;
  size_t words_el = 0;
  if (group == 3) PBC_DISPATCH_GT(P, words_el = G::WORDS_EL);
  else PBC_DISPATCH_G(P, group, words_el = F::WORDS_EL);
  // (the field-width dispatch is resolved here, BEFORE anything is allocated: the launches below repeat the same switch
  // and cannot take its failing default any more)
  const size_t units = group == 3 ? rows << kPpWin : rows * kPpRowLen;
  const size_t tab_bytes = units * words_el * 4 * (group == 3 ? 1 : 2);
  pbc_hip_element_pp_s *pp = new pbc_hip_element_pp_s{P, group, dev, nullptr, nullptr, false};
  DevBuf bflags;
  std::vector<uint8_t> flags(units);
  auto bail = [&](const char *what) {
    if (pp->tab) (void) hipFree(pp->tab);
    if (pp->base) (void) hipFree(pp->base);
    delete pp;
    return fail("element_pp_init: %s", what);
  };
  if (hipMalloc(&pp->tab, tab_bytes) != hipSuccess || hipMalloc(&pp->base, lrec) != hipSuccess || bflags.alloc(units) != hipSuccess)
    return bail("device allocation failed");
  if (hipMemcpy(pp->base, in, lrec, hipMemcpyHostToDevice) != hipSuccess) return bail("H2D copy failed");
  const unsigned grid = (unsigned) ((units + kBlock - 1) / kBlock);
  if (group == 3) {
    PBC_DISPATCH_GT(P, hipLaunchKernelGGL(gt_pp_init_kernel<G>, dim3(grid), dim3(kBlock), 0, 0, pp->tab, (const uint8_t *) pp->base, units, kargs<G::NW>(P)));
  } else {
    PBC_DISPATCH_G(P, group, hipLaunchKernelGGL(ec_pp_init_kernel<F>, dim3(grid), dim3(kBlock), 0, 0, pp->tab, bflags.as<uint8_t>(), (const uint8_t *) pp->base,
                                                P->len_zr, units, kargs<F::NW>(P)));
  }
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return bail("the table kernel failed");
  if (group != 3) {                    // (the kernel's status first: a faulted kernel is not a failed copy)
    if (hipMemcpy(flags.data(), bflags.p, units, hipMemcpyDeviceToHost) != hipSuccess) return bail("D2H copy failed");
    for (uint8_t f : flags) pp->complete_only |= f != 0;
  }
  *out = pp;
  return 0;
}
extern "C" void pbc_hip_element_pp_clear(pbc_hip_element_pp_t *pp) {
  if (!pp) return;
  DeviceGuard guard(pp->device);
  (void) hipFree(pp->tab);
  (void) hipFree(pp->base);
  delete pp;
}
static int pp_pow_launch(pbc_hip_element_pp_s *pp, void *d_out, const void *d_zr, size_t n, hipStream_t s, const OwnWs *own) {
  pbc_hip_pairing_s *P = pp->P;
  if (!n) return 0;
  uint8_t *o = (uint8_t *) d_out;
  const uint8_t *z = (const uint8_t *) d_zr;
  const unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock);
  if (pp->group == 3) {
    PBC_DISPATCH_GT(P, hipLaunchKernelGGL(gt_pp_pow_kernel<G>, dim3(grid), dim3(kBlock), 0, s, o, (const uint32_t *) pp->tab, z, P->len_zr, n, kargs<G::NW>(P)));
  } else if (pp->complete_only || P->group_slow) {
    PBC_DISPATCH_G(P, pp->group, hipLaunchKernelGGL(ec_mul_kernel<F>, dim3(grid), dim3(kBlock), 0, s, o, (const uint8_t *) pp->base, (size_t) 0, z, P->len_zr,
                                                    (const uint8_t *) nullptr, n, kargs<F::NW>(P)));
  } else {
    ProdWs W(P, s, own);
    uint8_t *flags = (uint8_t *) W.get(n);
    if (!flags) return 1;
    if (pp->group == 1 && P->nlimb == 5 && ((P->type == 'd' && P->deg == 3 && P->dconst.limb_ok) || (P->type == 'f' && P->fconst.pl_ok))) {
      if (P->type == 'd') hipLaunchKernelGGL(l5_pp_pow_kernel<KPd>, dim3(grid), dim3(kBlock), 0, s, o, (const uint32_t *) pp->tab, z, P->len_zr, flags, n, kargs<5>(P));
      else hipLaunchKernelGGL(l5_pp_pow_kernel<KPf>, dim3(grid), dim3(kBlock), 0, s, o, (const uint32_t *) pp->tab, z, P->len_zr, flags, n, kargs<5>(P));
      hipLaunchKernelGGL(ec_mul_kernel<FqOps<5>>, dim3(grid), dim3(kBlock), 0, s, o, (const uint8_t *) pp->base, (size_t) 0, z, P->len_zr, (const uint8_t *) flags, n, kargs<5>(P));
    } else {
      PBC_DISPATCH_G(P, pp->group, {
        hipLaunchKernelGGL(ec_pp_pow_kernel<F>, dim3(grid), dim3(kBlock), 0, s, o, (const uint32_t *) pp->tab, z, P->len_zr, flags, n, kargs<F::NW>(P));
        hipLaunchKernelGGL(ec_mul_kernel<F>, dim3(grid), dim3(kBlock), 0, s, o, (const uint8_t *) pp->base, (size_t) 0, z, P->len_zr, (const uint8_t *) flags, n, kargs<F::NW>(P));
      });
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
extern "C" int pbc_hip_element_pp_pow_zn_batch_dev(pbc_hip_element_pp_t *pp, void *d_out, const void *d_zr, size_t n, void *stream) {
  if (!pp) return fail("null pp");
  return pp_pow_launch(pp, d_out, d_zr, n, (hipStream_t) stream, nullptr);
}
extern "C" int pbc_hip_element_pp_pow_zn_batch(pbc_hip_element_pp_t *pp, uint8_t *out, const uint8_t *zr, size_t n) {
  if (!pp) return fail("null pp");
  pbc_hip_pairing_s *P = pp->P;
  const size_t lo = pp->group == 3 ? (size_t) P->lenT : pp->group == 2 ? (size_t) P->len2 : (size_t) P->len1;
  DeviceGuard guard(pp->device);
  if (P->ndev > 0) {
    // the object has a device set, the table lives on ONE device: this call runs there, staged (the set stays as it is
    // for the pairing entry points)
    if (!n) return 0;
    DevBuf bz, bo;
    HIP_TRY(bz.alloc(n * (size_t) P->len_zr));
    HIP_TRY(bo.alloc(n * lo));
    HIP_TRY(hipMemcpy(bz.p, zr, n * (size_t) P->len_zr, hipMemcpyHostToDevice));
    if (pp_pow_launch(pp, bo.p, bz.p, n, 0, nullptr)) return 1;
    HIP_TRY(hipMemcpy(out, bo.p, n * lo, hipMemcpyDeviceToHost));
    return 0;
  }
  return run_host_generic(P, out, lo, zr, (size_t) P->len_zr, nullptr, 0, n,
                          [pp](void *d_out, const void *d_a, const void *, size_t m, hipStream_t s, const OwnWs *own) {
                            return pp_pow_launch(pp, d_out, d_a, m, s, own);
                          }, false);
}
