// pairing_fw.cuh -- Type F (BN, k = 12) on the five-word fields, ONE PAIRING PER WAVEFRONT (round 6; small batches).
//
// Same value as f_pairing (ecc/f_param.c:289-311: cc_miller_no_denom :216-233 + f_tateexp :250-283), the formulas of
// pairing_f.cuh.  The throughput kernel runs a pairing as one lane's instruction stream: 7.0 ms through the reference's call
// sites whatever the batch size (one CPU core: 6.6 ms).  Here a wavefront owns one pairing on the machine of pairing_dw.cuh:
// every F_q element a slot of an LDS slot file (218 slots), a LEVEL = every lane computes one lazily reduced sum of F_q products
// from the slots its table row names (fw_tables.h, generated and checked against the reference's vectors on Python integers by
// tools/fw_gen.py), the schedule a straight line the host writes once per object (fw_sched.h; 1306 levels):
//   * V <- 2V / V <- V +- P on E(F_q) with the line's values a Qx', a Qx' xi, b Qy', b Qy' xi (x, y, beta y) in its last level;
//   * the accumulator times the line in ONE level (twelve sums of five terms, four lanes each), its square in two (the doubled,
//     beta- and xi-scaled copies, then twelve sums of six or eight terms) -- which also carry the first two levels of the NEXT
//     step's doubling;
//   * general F_q^12 products of the final exponentiation in three levels, Frobenius maps / conjugations / copies in one,
//     the one inversion (norm of polymod_invert down to F_q) as lane code.
// The set-up (byte loads, curve checks, the untwisting map, the powers of X^(q^2) / X and X^q / X) is ordinary code on lane 0.
#pragma once
#include "pairing_dw.cuh"
#include "fw_tables.h"

namespace pbc {

struct FwTables {
  static constexpr int kSlots = fw::kSlots, kRows = fw::kRows;
  static PBC_DEV const uint32_t *rows_src() { return fw::g_rows; }
  static PBC_DEV int line_slot(int) { return 0; }                  // (type f has no pairing_pp tables)
};

template <int ND>
struct FW : DW<ND, FwTables> {
  typedef DW<ND, FwTables> VM;
  typedef TypeF<ND> F;
  typedef typename F::fq fq;
  typedef typename F::g2 g2;
  using VM::put_fq; using VM::get_fq; using VM::slot; using VM::uniform64; using VM::run_entry; using VM::begin;
  static constexpr int L = VM::L;

  static PBC_DEV void put_g2(int s, const g2 &a, const fq &beta) {   // x, y, beta y in three consecutive slots
    fq by;
    fp_mul<ND>(by, a.y, beta);
    put_fq(s, a.x); put_fq(s + 1, a.y); put_fq(s + 2, by);
  }
  // ---- lane 0: constants, bytes -> slots, curve checks, the untwisting map (f_miller_lane) ----
  static __device__ __noinline__ bool setup(const uint8_t *g1, const uint8_t *g2b) {
    using namespace fw;
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
    const fq beta = F::dk(c_f.beta);
    put_fq(S_BETA, beta);
    fp_neg<ND>(t, beta); put_fq(S_NBETA, t);
    const g2 na = F::fk2(c_f.negalpha);
    put_fq(S_NAX, na.x); put_fq(S_NAY, na.y);
    fp_mul<ND>(t, na.x, beta); put_fq(S_BNAX, t);
    fp_mul<ND>(t, na.y, beta); put_fq(S_BNAY, t);
    {
      // X^(q^2) = e2 X (xpowq2), X^q = gamma X: their powers
      const g2 e2 = F::fk2(c_f.xpowq2), gm = F::fk2(c_f.gamma);
      g2 ep = e2, gp = gm;
      for (int i = 1; i < 6; i++) {
        put_g2(S_E2_1_x + 3 * (i - 1), ep, beta);
        fq nby, nx;
        fp_mul<ND>(nby, gp.y, beta);
        fp_neg<ND>(nby, nby);
        fp_neg<ND>(nx, gp.x);
        put_fq(S_G_1_x + 4 * (i - 1), gp.x); put_fq(S_G_1_x + 4 * (i - 1) + 1, gp.y); put_fq(S_G_1_x + 4 * (i - 1) + 2, nby); put_fq(S_G_1_x + 4 * (i - 1) + 3, nx);
        g2 n;
        F::g2_mul(n, ep, e2); ep = n;
        F::g2_mul(n, gp, gm); gp = n;
      }
    }
    // inputs
    fq Px, Py;
    g2 Qx, Qy;
    fp_load_be<ND>(Px, g1);
    fp_load_be<ND>(Py, g1 + NB);
    F::g2_load_be(Qx, g2b);
    F::g2_load_be(Qy, g2b + 2 * NB);
    bool valid;
    {
      // curve_is_valid_point (curve.c:57-77): E: y^2 = x^3 + b;  E': y^2 = x^3 - alpha b over F_q^2
      fq t0, t1;
      fp_sqr<ND>(t0, Px);
      fp_mul<ND>(t0, t0, Px);
      fp_add<ND>(t0, t0, F::dk(c_f.B));
      fp_sqr<ND>(t1, Py);
      valid = fp_eq<ND>(t0, t1);
      g2 u0, u1;
      F::g2_sqr(u0, Qx);
      F::g2_mul(u0, u0, Qx);
      F::g2_add(u0, u0, F::fk2(c_f.tb));
      F::g2_sqr(u1, Qy);
      valid &= F::g2_eq(u0, u1);
    }
    // untwist: (x, y) -> (x negalphainv X^4, y negalphainv X^3)  (f_pairing, f_param.c:296-303); the copies times xi = X^6 are Q itself
    {
      const g2 ni = F::fk2(c_f.negalphainv);
      g2 qa, qb;
      F::g2_mul(qa, Qx, ni);
      F::g2_mul(qb, Qy, ni);
      put_g2(S_QA_x, qa, beta); put_g2(S_QAN_x, Qx, beta); put_g2(S_QB_x, qb, beta); put_g2(S_QBN_x, Qy, beta);
    }
    put_fq(S_X, Px); put_fq(S_Y, Py); put_fq(S_Z, one); put_fq(S_ZZ, one); put_fq(S_ZZZ, one);
    fp_neg<ND>(t, one); put_fq(S_nZ, t);
    put_fq(S_Px, Px); put_fq(S_Py, Py);
    fp_neg<ND>(t, Py); put_fq(S_nPy, t);
    put_fq(S_F_0_x, one);                             // F = 1 (the other eleven slots are zero from begin())
    return valid;
  }
  static PBC_DEV void inversion() {
    if (threadIdx.x == 0) {
      fq n;
      fp_inv<ND>(n, get_fq(fw::S_nrm));               // the only inversion
      put_fq(fw::S_ninv, n);
    }
    __builtin_amdgcn_wave_barrier();
  }
  static __device__ __noinline__ void interpret(const uint64_t *sched_) {
    const uint64_t *sched = reinterpret_cast<const uint64_t *>(uniform64(reinterpret_cast<uint64_t>(sched_)));
    uint64_t e = uniform64(sched[0]);
    for (int k = 1;; k++) {
      const int op = (int) ((e >> 38) & 15u);
      if (op == fw::OP_END) break;
      const uint64_t nxt = uniform64(sched[k]);       // (a scalar load that completes under this entry's work)
      if (op == fw::OP_LEVEL) run_entry(e, nullptr);
      else inversion();
      e = nxt;
    }
  }
  static PBC_DEV void store_gt(uint8_t *gt, bool valid) {
    if (threadIdx.x < 12) {
      fq o = get_fq(fw::S_F_0_x + (int) threadIdx.x);   // F.0.x, F.0.y, F.1.x, ...: consecutive slots in GT's wire order
      if (!valid) {                                   // an input that deserialises to O: the identity of GT
        fq one;
        fp_set<ND>(one, fpk<ND>().one);
#pragma unroll
        for (int k = 0; k < ND; k++) o.v[k] = threadIdx.x == 0 ? one.v[k] : 0u;
      }
      fp_store_be<ND>(gt + (size_t) threadIdx.x * fpk<ND>().fbytes, o);
    }
  }
  // element_prod_pairing (type f installs no product routine: generic_prod_pairings, ecc/pairing.c:35-46, multiplies k reduced
  // pairings; the reduced product of the Miller values is the same element).  First kernel: the Miller value of ONE TERM -> its
  // record of the workspace (kRec words: the twelve slots of F as they are, then the validity flag).
  static constexpr int kRec = (12 * L + 1 + 7) & ~7;
  static __device__ void miller_term(uint32_t *rec, const uint8_t *g1, const uint8_t *g2b, const uint64_t *sched) {
    begin();
    __shared__ int valid_s;
    if (threadIdx.x == 0) valid_s = setup(g1, g2b) ? 1 : 0;
    __builtin_amdgcn_wave_barrier();
    interpret(sched);
    for (int i = (int) threadIdx.x; i < 12 * L; i += 64) rec[i] = slot(fw::S_F_0_x)[i];
    if (threadIdx.x == 0) rec[12 * L] = (uint32_t) valid_s;
  }
  // second kernel: F <- the product of the k records (register U takes each further one, the three levels of mul_F_F_U are
  // sched[0..2]), then the final exponentiation (sched + 3); any invalid term: the identity
  static __device__ void finish(uint8_t *gt, const uint32_t *recs, int k, const uint64_t *sched_) {
    const uint64_t *sched = reinterpret_cast<const uint64_t *>(uniform64(reinterpret_cast<uint64_t>(sched_)));
    begin();
    __shared__ int dummy_s;
    if (threadIdx.x == 0) {                           // the constants only: a set-up on records of zeros leaves F = 1, replaced below
      __attribute__((aligned(4))) uint8_t z1[8 * ND], z2[16 * ND];
      for (int i = 0; i < 8 * ND; i++) z1[i] = 0;
      for (int i = 0; i < 16 * ND; i++) z2[i] = 0;
      dummy_s = setup(z1, z2) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    bool valid = recs[12 * L] != 0;
    for (int i = (int) threadIdx.x; i < 12 * L; i += 64) slot(fw::S_F_0_x)[i] = recs[i];
    const uint64_t e0 = uniform64(sched[0]), e1 = uniform64(sched[1]), e2 = uniform64(sched[2]);
    for (int t = 1; t < k; t++) {
      const uint32_t *r = recs + (size_t) t * kRec;
      valid &= r[12 * L] != 0;
      for (int i = (int) threadIdx.x; i < 12 * L; i += 64) slot(fw::S_U_0_x)[i] = r[i];
      __builtin_amdgcn_wave_barrier();
      run_entry(e0, nullptr);
      run_entry(e1, nullptr);
      run_entry(e2, nullptr);
    }
    __builtin_amdgcn_wave_barrier();
    interpret(sched + 3);
    store_gt(gt, valid);
  }
  // element_pairing
  static __device__ void pairing(uint8_t *gt, const uint8_t *g1, const uint8_t *g2b, const uint64_t *sched) {
    begin();
    __shared__ int valid_s;
    if (threadIdx.x == 0) valid_s = setup(g1, g2b) ? 1 : 0;
    __builtin_amdgcn_wave_barrier();
    interpret(sched);
    store_gt(gt, valid_s != 0);
  }
};

}  // namespace pbc
