// pairing_gw.cuh -- Type G (MNT, k = 10) on the five-word fields, ONE PAIRING PER WAVEFRONT (round 6; small batches).
//
// Same value as the type g pairing (ecc/g_param.c: the Miller loop of cc_miller_no_denom_affine and cc_tatepower :471-558), the
// formulas of pairing_d.cuh's TypeMNT<5, 5>.  The throughput kernel runs a pairing as one lane's instruction stream: 14.4 ms through
// the reference's call sites whatever the batch size (one CPU core: 4.3 ms).  Here a wavefront owns one pairing on the machine
// of pairing_dw.cuh: every F_q element a slot of an LDS slot file (240 slots), a LEVEL = every lane computes one lazily reduced
// sum of F_q products from the slots its table row names (gw_tables.h, generated and checked against the reference's vectors on
// Python integers by tools/gw_gen.py: sums of more than eight terms are chained, over-full levels repacked), the schedule a straight
// line the host writes once per object (gw_sched.h; 2542 levels).  The set-up (byte loads, curve checks, twist map), the B = 0
// test, the one inversion and the store are ordinary code on lane 0 / lanes 0-9.
#pragma once
#include "pairing_dw.cuh"
#include "gw_tables.h"

namespace pbc {

struct GwTables {
  static constexpr int kSlots = gw::kSlots, kRows = gw::kRows;
  static PBC_DEV const uint32_t *rows_src() { return gw::g_rows; }
  static PBC_DEV int line_slot(int) { return gw::S_La; }           // La, Lb, Lc: consecutive slots
};

template <int ND>
struct GW : DW<ND, GwTables> {
  typedef DW<ND, GwTables> VM;
  static constexpr int DEG = 5;
  typedef TypeMNT<ND, DEG> G;
  typedef typename G::fq fq;
  typedef typename G::f3 f5;                           // (pairing_d.cuh keeps type d's names: f3 = F_q^d)
  using VM::put_fq; using VM::get_fq; using VM::put; using VM::slot; using VM::uniform64; using VM::run_entry; using VM::begin;
  using VM::line_fetch; using VM::line_put;
  static constexpr int L = VM::L;

  // ---- lane 0: constants, bytes -> slots, curve checks, twist map (d_setup_lane) ----
  // (g1 == nullptr: pairing_pp_apply -- no first argument, the table stands for it)
  static __device__ __noinline__ bool setup(const uint8_t *g1, const uint8_t *g2) {
    using namespace gw;
    const FpK<ND> &K = fpk<ND>();
    const int NB = (int) K.fbytes;
    fq one, zero, t, u;
    fp_set<ND>(one, K.one);
#pragma unroll
    for (int k = 0; k < ND; k++) zero.v[k] = 0;
    put_fq(S_ZERO, zero);
    put_fq(S_ONE, one);
    fp_neg<ND>(t, one); put_fq(S_M1, t);
    fp_dbl<ND>(t, one); put_fq(S_TWO, t);
    fp_neg<ND>(u, t); put_fq(S_M2, u);
    fp_add<ND>(u, t, one); put_fq(S_THREE, u);
    fp_dbl<ND>(t, t); put_fq(S_FOUR, t);
    fp_dbl<ND>(t, t); fp_neg<ND>(u, t); put_fq(S_M8, u);
    fp_dbl<ND>(t, t); put_fq(S_SIXTEEN, t);
    fp_halve<ND>(t, one); put_fq(S_HALF, t);
    fp_halve<ND>(t, G::dk(c_d.nqrinv)); fp_halve<ND>(t, t); put_fq(S_QVI, t);
    put_fq(S_A, G::dk(c_d.A));
    const fq v = G::dk(c_d.nqr);
    put_fq(S_V, v);
    fp_neg<ND>(t, v); put_fq(S_NV, t);
    for (int j = 0; j < DEG - 1; j++)
      for (int k = 0; k < DEG; k++) {
        fl<ND> a;
        for (int l = 0; l < L; l++) a.l[l] = c_d.xpwr29[(j * DEG + k) * L + l];
        put(S_XP5_0 + j * DEG + k, a);
        const fq xq = G::dk(c_d.xpowq[j][k]);
        put_fq(S_XQ1_0 + j * DEG + k, xq);
        fp_neg<ND>(t, xq);
        put_fq(S_NXQ1_0 + j * DEG + k, t);
      }
    // inputs
    fq Px = zero, Py = zero;
    f5 Qx, Qy;
    if (g1) {
      fp_load_be<ND>(Px, g1);
      fp_load_be<ND>(Py, g1 + NB);
    }
    G::f3_load_be(Qx, g2);
    G::f3_load_be(Qy, g2 + DEG * NB);
    bool valid;
    {
      // curve_is_valid_point (curve.c:57-77): E: y^2 = x^3 + a x + b; the twist over F_q^5
      fq t0, t1;
      fp_sqr<ND>(t0, Px);
      fp_add<ND>(t0, t0, G::dk(c_d.A));
      fp_mul<ND>(t0, t0, Px);
      fp_add<ND>(t0, t0, G::dk(c_d.B));
      fp_sqr<ND>(t1, Py);
      valid = fp_eq<ND>(t0, t1) | (g1 == nullptr);
      f5 u0, u1;
      G::f3_sqr(u0, Qx);
      fp_add<ND>(u0.c[0], u0.c[0], G::dk(c_d.ta));
      G::f3_mul(u0, u0, Qx);
      fp_add<ND>(u0.c[0], u0.c[0], G::dk(c_d.tb));
      G::f3_sqr(u1, Qy);
      valid &= G::f3_eq(u0, u1);
    }
    // twist map (x, y) -> (v^-1 x, v^-2 y sqrt(v)); v times the second: v^-1 y
    {
      f5 qx, qy, vqy;
      G::f3_mul_fq(qx, Qx, G::dk(c_d.nqrinv));
      G::f3_mul_fq(qy, Qy, G::dk(c_d.nqrinv2));
      G::f3_mul_fq(vqy, Qy, G::dk(c_d.nqrinv));
      for (int i = 0; i < DEG; i++) { put_fq(S_Qx0 + i, qx.c[i]); put_fq(S_Qy0 + i, qy.c[i]); put_fq(S_VQy0 + i, vqy.c[i]); }
    }
    put_fq(S_X, Px); put_fq(S_Y, Py); put_fq(S_Z, one); put_fq(S_ZZ, one); put_fq(S_ZZZ, one);
    fp_neg<ND>(t, one); put_fq(S_nZ, t);
    put_fq(S_W, G::dk(c_d.A));
    put_fq(S_Px, Px); put_fq(S_Py, Py);
    fp_neg<ND>(t, Py); put_fq(S_nPy, t);
    put_fq(S_f_x0, one);                               // f = 1 (the other nine slots are zero from begin())
    return valid;
  }
  static PBC_DEV void bzero_test() {
    using namespace gw;
    if (threadIdx.x == 0) {
      // B = 0 (the value after the easy part is +-1) must not poison 1 / (D B): invert D * 1 instead (d_final_exp)
      fq one, zero;
      fp_set<ND>(one, fpk<ND>().one);
#pragma unroll
      for (int k = 0; k < ND; k++) zero.v[k] = 0;
      fq b[DEG];
      bool b0 = true;
      for (int i = 0; i < DEG; i++) { b[i] = get_fq(S_wB0 + i); b0 &= fp_is0<ND>(b[i]); }
      for (int i = 0; i < DEG; i++) put_fq(S_Bn0 + i, b0 ? (i ? zero : one) : b[i]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  static PBC_DEV void inversion() {
    if (threadIdx.x == 0) {
      fq n;
      fp_inv<ND>(n, get_fq(gw::S_nrm));                // the only inversion
      put_fq(gw::S_ninv, n);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // runs entries up to OP_END or OP_MARK; returns how many it consumed (the terminator included)
  static __device__ __noinline__ int interpret(const uint64_t *sched_, const uint32_t *tab_ = nullptr) {
    const uint64_t *sched = reinterpret_cast<const uint64_t *>(uniform64(reinterpret_cast<uint64_t>(sched_)));
    const uint32_t *tab = reinterpret_cast<const uint32_t *>(uniform64(reinterpret_cast<uint64_t>(tab_)));
    uint64_t e = uniform64(sched[0]);
    int k = 1;
    for (;; k++) {
      const int op = (int) ((e >> 38) & 15u);
      if (op == gw::OP_END || op == gw::OP_MARK) break;
      const uint64_t nxt = uniform64(sched[k]);        // (a scalar load that completes under this entry's work)
      if (op == gw::OP_LEVEL) run_entry(e, tab);
      else if (op == gw::OP_BZERO) bzero_test();
      else if (op == gw::OP_INV) inversion();
      else { const int line = (int) ((e >> 42) & 4095u); line_put(line_fetch(tab, line), line); }
      e = nxt;
    }
    return __builtin_amdgcn_readfirstlane(k);
  }
  static PBC_DEV void store_gt(uint8_t *gt, bool valid) {
    if (threadIdx.x < 2 * DEG) {
      fq o = get_fq(gw::S_f_x0 + (int) threadIdx.x);   // f.x0..4, f.y0..4 are consecutive slots: GT's wire order
      if (!valid) {                                    // an input that deserialises to O: the identity of GT
        fq one;
        fp_set<ND>(one, fpk<ND>().one);
#pragma unroll
        for (int k = 0; k < ND; k++) o.v[k] = threadIdx.x == 0 ? one.v[k] : 0u;
      }
      fp_store_be<ND>(gt + (size_t) threadIdx.x * fpk<ND>().fbytes, o);
    }
  }
  // element_prod_pairing, first kernel: the Miller value of ONE TERM -> its record (kRec words: the ten slots of f, the validity flag)
  static constexpr int kRec = (2 * DEG * L + 1 + 7) & ~7;
  static __device__ void miller_term(uint32_t *rec, const uint8_t *g1, const uint8_t *g2, const uint64_t *sched) {
    begin();
    __shared__ int valid_s;
    if (threadIdx.x == 0) valid_s = setup(g1, g2) ? 1 : 0;
    __builtin_amdgcn_wave_barrier();
    interpret(sched);
    if (threadIdx.x < 2 * DEG * L) rec[threadIdx.x] = slot(gw::S_f_x0)[threadIdx.x];
    if (threadIdx.x == 0) rec[2 * DEG * L] = (uint32_t) valid_s;
  }
  // second kernel: f <- the product of the k records (register u takes each further one; the levels of mul_f_u are the schedule's
  // entries up to its mark), then the final exponentiation; any invalid term: the identity
  static __device__ void finish(uint8_t *gt, const uint32_t *recs, int k, const uint64_t *sched) {
    begin();
    __shared__ int dummy_s;
    if (threadIdx.x == 0) {                            // the constants only: a set-up on a record of zeros, f replaced below
      __attribute__((aligned(4))) uint8_t z2[8 * DEG * ND];
      for (int i = 0; i < 8 * DEG * ND; i++) z2[i] = 0;
      dummy_s = setup(nullptr, z2) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    bool valid = recs[2 * DEG * L] != 0;
    if (threadIdx.x < 2 * DEG * L) slot(gw::S_f_x0)[threadIdx.x] = recs[threadIdx.x];
    int used = 0;
    for (int t = 1; t < k; t++) {
      const uint32_t *r = recs + (size_t) t * kRec;
      valid &= r[2 * DEG * L] != 0;
      if (threadIdx.x < 2 * DEG * L) slot(gw::S_u_x0)[threadIdx.x] = r[threadIdx.x];
      __builtin_amdgcn_wave_barrier();
      used = interpret(sched);
    }
    if (k < 2) {                                       // (never routed here; skip the product's entries)
      for (used = 0; ((uniform64(sched[used]) >> 38) & 15u) != (uint64_t) gw::OP_MARK; used++) {}
      used++;
    }
    __builtin_amdgcn_wave_barrier();
    interpret(sched + used);
    store_gt(gt, valid);
  }
  // pairing_pp_apply: the lines' coefficients come from the table of pairing_pp_init (pairing_d.cuh d_pp_init_lane)
  static __device__ void pp_apply(uint8_t *gt, const uint32_t *tab, bool p_valid, const uint8_t *g2, const uint64_t *sched) {
    begin();
    __shared__ int valid_s;
    if (threadIdx.x == 0) valid_s = setup(nullptr, g2) ? 1 : 0;
    __builtin_amdgcn_wave_barrier();
    interpret(sched, tab);
    store_gt(gt, p_valid && valid_s != 0);
  }
  // element_pairing
  static __device__ void pairing(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, const uint64_t *sched) {
    begin();
    __shared__ int valid_s;
    if (threadIdx.x == 0) valid_s = setup(g1, g2) ? 1 : 0;
    __builtin_amdgcn_wave_barrier();
    interpret(sched);
    store_gt(gt, valid_s != 0);
  }
};

}  // namespace pbc
