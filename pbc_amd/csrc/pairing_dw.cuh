// pairing_dw.cuh -- Type D (MNT, k = 6) on the five-word fields, ONE PAIRING PER WAVEFRONT (round 6; small batches).
//
// Same value as cc_pairing (ecc/d_param.c:570-587: cc_miller_no_denom_affine :321-422 + cc_tatepower :505-564), the
// formulas of pairing_d.cuh.  The throughput kernel runs a pairing as one lane's serial instruction stream: 2.0 M vector
// instructions, 3.9 ms through the reference's call sites whatever the batch size -- three times what one CPU core needs
// (VERDICT r5 "missing" 1).  Every tower operation, though, is a set of INDEPENDENT lazily reduced sums of F_q products.
// Here a wavefront owns one pairing:
//   * every F_q element is a slot of six 29-bit limbs in an LDS slot file (149 slots, 3.5 KB per wavefront);
//   * a LEVEL is "lane l computes out[l] = sum_t x[l][t] y[l][t] / R" -- the library's sop_limbs on operands gathered from
//     the slot file by the lane's ROW of a table (dw_tables.h, generated and checked against the reference's vectors on
//     Python integers by tools/dw_gen.py) -- "and writes its slot"; a level's lanes read before any lane writes;
//   * sums, differences and small multiples are products with the constant slots 1, -1, 2, ...: nothing but sums of
//     products exists, so ONE routine (2 / 4 terms per lane; eight-term sums on four lanes each) is the whole arithmetic;
//   * the first lanes run the ACCUMULATOR track (f <- f * line and f <- f^2: 2 levels each), the last lanes the POINT track
//     (V <- 2V in modified Jacobian coordinates: 4 levels, V <- V +- P: 5; the line's value at Q rides in the next point
//     program) ahead of it, each advancing one level per step of the machine; the final exponentiation (Lucas ladder: 2
//     levels per exponent bit) runs on one track.
// 1262 levels per pairing (the throughput kernel: ~30 000 dependent products).  Which level of which program runs when
// depends on the curve's constants only: the HOST flattens a pairing into a schedule (dw_sched.h) and this kernel interprets
// it.  The set-up (byte loads, curve checks, twist map), the B = 0 test, the one inversion and the store are ordinary lane
// code on lane 0 / lanes 0-5.  Measured (profiles/r06_notes.md): one pairing 1.15 ms (lane kernel 4.1 ms, one CPU core 1.35 ms),
// up to 1024 pairings at the latency of one, 1.35 M pairings/s at 4096.
#pragma once
#include "pairing_d.cuh"
#include "dw_tables.h"

#ifndef PBC_DW_WHATIF
#define PBC_DW_WHATIF 0               // timing what-ifs (wrong results): bit 1 no Miller loop, bit 2 no final exponentiation
#endif
namespace pbc {

// (the machine does not know the pairing: TB names the tables -- slots, rows -- it runs; pairing_fw.cuh runs type f's on it)
struct DwTables {
  static constexpr int kSlots = dw::kSlots, kRows = dw::kRows;
  static PBC_DEV const uint32_t *rows_src() { return dw::g_rows; }
  static PBC_DEV int line_slot(int line) { return dw::S_L0_a + 3 * (line & 1); }   // where a', b', c' of a pairing_pp table line go
};
template <int ND, class TB> __shared__ __attribute__((aligned(16))) uint32_t g_lds_wv[TB::kSlots * Limbs29<ND>::L + TB::kRows * 5];

template <int ND, class TB = DwTables>
struct DW {
  typedef TypeMNT<ND, 3> D;
  typedef typename D::fq fq;
  typedef typename D::f3 f3;
  static constexpr int L = Limbs29<ND>::L;
  // (the row tables hold sums of up to eight terms.  Six and seven limbs: eight products of every limb pair and the reduction's
  // fit a 64-bit column, fp.cuh sop_limbs.  Eight limbs: the eight terms alone do -- 64 products below 2^58 --, and the four
  // lanes' joined columns are relieved exactly, wide_squeeze, before the reduction adds its own: exec_split8.)
  static constexpr bool kSqueeze = 8 * L + L > 63;
  static_assert(4 * L + L <= 63 && 8 * L <= 64, "sums of four terms on a lane, of eight on four lanes");

  static PBC_DEV uint32_t *slot(int s) { return g_lds_wv<ND, TB> + s * L; }
  static PBC_DEV const uint32_t *rows() { return g_lds_wv<ND, TB> + TB::kSlots * L; }
  static PBC_DEV void put(int s, const fl<ND> &a) {
#pragma unroll
    for (int i = 0; i < L; i++) slot(s)[i] = a.l[i];
  }
  static PBC_DEV fl<ND> get(int s) {
    fl<ND> r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = slot(s)[i];
    return r;
  }
  static PBC_DEV void put_fq(int s, const fq &a) { fl<ND> t; to_limbs<ND>(t, a); put(s, t); }
  static PBC_DEV fq get_fq(int s) { fq r; from_limbs<ND>(r, get(s)); return r; }     // the canonical residue (Montgomery form)

  struct Lev { int row, T, lanes; };
  // a schedule entry (dw_sched.h): row a [0:12) | lanes a [12:17) | row b [17:29) | lanes b [29:34) | terms [34:38) | op [38:42)
  static PBC_DEV Lev ent_a(uint64_t e) { return Lev{(int) (e & 4095u), (int) ((e >> 34) & 15u), (int) ((e >> 12) & 31u)}; }
  static PBC_DEV Lev ent_b(uint64_t e) { return Lev{(int) ((e >> 17) & 4095u), (int) ((e >> 34) & 15u), (int) ((e >> 29) & 31u)}; }

  // One level of the machine: track a on lanes 0-31, track b on lanes 32-63 (lanes = 0: the track idles).  A lane's ROW
  // (the slot it writes, eight x and eight y operand slots) does not depend on data: the rows of level k + 1 are read while
  // level k multiplies (one LDS round trip less on every level's critical path).
  struct Rows { uint32_t w[5]; bool active; };
  static PBC_DEV Rows load_rows(const Lev a, const Lev b) {
    const int lane = (int) threadIdx.x, r = lane & 31;
    const bool second = lane >= 32;
    const int first = second ? b.row : a.row, lanes = second ? b.lanes : a.lanes;
    Rows R;
    R.active = r < lanes;
    const uint32_t *p = rows() + (first + (R.active ? r : 0)) * 5;
#pragma unroll
    for (int i = 0; i < 5; i++) R.w[i] = R.active ? p[i] : 0u;           // (slot 0 is ZERO: an idle lane multiplies zeros)
    return R;
  }
  template <int T>
  static PBC_DEV void exec_T(const Rows &R) {
    fl<ND> x[T], y[T], res;
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int kx = 1 + t, ky = 9 + t;
      const uint32_t *px = slot((int) ((R.w[kx >> 2] >> (8 * (kx & 3))) & 255u));
      const uint32_t *py = slot((int) ((R.w[ky >> 2] >> (8 * (ky & 3))) & 255u));
#pragma unroll
      for (int i = 0; i < L; i++) { x[t].l[i] = px[i]; y[t].l[i] = py[i]; }
    }
    sop_limbs<ND, T>(res, x, y);
    __builtin_amdgcn_wave_barrier();                                      // every lane has read: now the writes
    if (R.active) put((int) (R.w[0] & 255u), res);
    __builtin_amdgcn_wave_barrier();
  }
  // Eight-term levels, FOUR lanes per sum (PBC_DW_SPLIT): a lone wavefront gets one v_mad_u64_u32 through every ~9 cycles, so
  // a level's time is its multiply-adds per LANE.  Lane s of a group of four accumulates terms s and s + 4 unreduced (wide
  // columns), two DPP exchanges (quad_perm 1032 / 2301) add the four partial columns, every lane reduces, lane 0 of the
  // group writes: 72 + 36 multiply-adds per lane instead of 288 + 36.
  static PBC_DEV Rows load_rows4(const Lev a, const Lev b) {
    const int g = (int) threadIdx.x >> 2;
    const bool second = g >= a.lanes;               // track a's sums take the first groups, track b's the next (at most 16 in all: dw_gen.py)
    const int r = second ? g - a.lanes : g, first = second ? b.row : a.row, lanes = second ? b.lanes : a.lanes;
    Rows R;
    R.active = r < lanes;
    const uint32_t *p = rows() + (first + (R.active ? r : 0)) * 5;
#pragma unroll
    for (int i = 0; i < 5; i++) R.w[i] = R.active ? p[i] : 0u;
    return R;
  }
  static PBC_DEV uint32_t row_byte(const Rows &R, int k) {                // k wave-uniform or per lane
    uint32_t w = R.w[0];
#pragma unroll
    for (int i = 1; i < 5; i++) w = (k >> 2) == i ? R.w[i] : w;
    return (w >> (8 * (k & 3))) & 255u;
  }
  static PBC_DEV void exec_split8(const Rows &R) {
    const int s = (int) threadIdx.x & 3;
    wide<ND> W;
    wide_zero<ND>(W);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int t = s + 4 * j;
      const uint32_t *px = slot((int) row_byte(R, 1 + t)), *py = slot((int) row_byte(R, 9 + t));
      fl<ND> x, y;
#pragma unroll
      for (int i = 0; i < L; i++) { x.l[i] = px[i]; y.l[i] = py[i]; }
      wide_mac<ND>(W, x, y);
    }
#pragma unroll
    for (int k = 0; k < 2 * L - 1; k++) {
      uint32_t lo = (uint32_t) W.c[k], hi = (uint32_t) (W.c[k] >> 32);
      uint32_t plo = (uint32_t) __builtin_amdgcn_mov_dpp((int) lo, 0xB1, 0xF, 0xF, false), phi = (uint32_t) __builtin_amdgcn_mov_dpp((int) hi, 0xB1, 0xF, 0xF, false);
      W.c[k] += ((uint64_t) phi << 32) | plo;                             // + the neighbour's (lanes 0<->1, 2<->3)
      lo = (uint32_t) W.c[k]; hi = (uint32_t) (W.c[k] >> 32);
      plo = (uint32_t) __builtin_amdgcn_mov_dpp((int) lo, 0x4E, 0xF, 0xF, false); phi = (uint32_t) __builtin_amdgcn_mov_dpp((int) hi, 0x4E, 0xF, 0xF, false);
      W.c[k] += ((uint64_t) phi << 32) | plo;                             // + the other pair's (0<->2, 1<->3)
    }
    fl<ND> res;
    if constexpr (kSqueeze) wide_squeeze<ND>(W);
    wide_reduce<ND>(res, W);
    __builtin_amdgcn_wave_barrier();
    if (R.active && s == 0) put((int) (R.w[0] & 255u), res);
    __builtin_amdgcn_wave_barrier();
  }
#ifndef PBC_DW_SPLIT
#define PBC_DW_SPLIT 1
#endif
  static PBC_DEV void exec(const Rows &R, const Rows &R4, int T) {
    if (T <= 2) exec_T<2>(R);
    else if (T <= 4) exec_T<4>(R);
    else if constexpr (PBC_DW_SPLIT || kSqueeze) exec_split8(R4);
    else exec_T<8>(R);
  }
  // ---- lane 0: constants; bytes -> slots, curve checks, twist map (d_setup_lane) ----
  static __device__ __noinline__ void setup_consts() {
    using namespace dw;
    const FpK<ND> &K = fpk<ND>();
    fq one, t, u;
    fp_set<ND>(one, K.one);
    fq zero;
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
    fp_halve<ND>(t, D::dk(c_d.nqrinv)); fp_halve<ND>(t, t); put_fq(S_QVI, t);
    put_fq(S_A, D::dk(c_d.A));
    put_fq(S_V, D::dk(c_d.nqr));
    for (int k = 0; k < 3; k++) {
      fl<ND> a, b;
      for (int l = 0; l < L; l++) { a.l[l] = c_d.xpwr29[(0 * 3 + k) * L + l]; b.l[l] = c_d.xpwr29[(1 * 3 + k) * L + l]; }
      put(S_XP3_0 + k, a);
      put(S_XP4_0 + k, b);
      put_fq(S_XQ1_0 + k, D::dk(c_d.xpowq[0][k]));
      put_fq(S_XQ2_0 + k, D::dk(c_d.xpowq[1][k]));
      fq xa, xb;                                      // v x^3, v x^4: the folds of v ay by in the two-level F_q^6 products
      from_limbs<ND>(xa, a);
      from_limbs<ND>(xb, b);
      fp_mul<ND>(xa, xa, D::dk(c_d.nqr));
      fp_mul<ND>(xb, xb, D::dk(c_d.nqr));
      put_fq(S_VXP3_0 + k, xa);
      put_fq(S_VXP4_0 + k, xb);
    }
    for (int i = 0; i < 3; i++) { put_fq(S_f_x0 + i, i ? zero : one); put_fq(S_f_y0 + i, zero); }     // f = 1
  }
  // the first argument: curve_is_valid_point (curve.c:57-77) on E: y^2 = x^3 + a x + b, then the point track's state
  static __device__ __noinline__ bool setup_point(const uint8_t *g1) {
    using namespace dw;
    const int NB = (int) fpk<ND>().fbytes;
    fq one, t, Px, Py, t0, t1;
    fp_set<ND>(one, fpk<ND>().one);
    fp_load_be<ND>(Px, g1);
    fp_load_be<ND>(Py, g1 + NB);
    fp_sqr<ND>(t0, Px);
    fp_add<ND>(t0, t0, D::dk(c_d.A));
    fp_mul<ND>(t0, t0, Px);
    fp_add<ND>(t0, t0, D::dk(c_d.B));
    fp_sqr<ND>(t1, Py);
    put_fq(S_X, Px); put_fq(S_Y, Py); put_fq(S_Z, one); put_fq(S_ZZ, one); put_fq(S_ZZZ, one);
    fp_neg<ND>(t, one); put_fq(S_nZ, t);
    put_fq(S_W, D::dk(c_d.A));
    put_fq(S_Px, Px); put_fq(S_Py, Py);
    fp_neg<ND>(t, Py); put_fq(S_nPy, t);
    return fp_eq<ND>(t0, t1);
  }
  // the second argument: the check on the twist over F_q^3, then the twist map (x, y) -> (v^-1 x, v^-2 y sqrt(v))
  // (cc_pairing, d_param.c:580-582)
  static __device__ __noinline__ bool setup_twist(const uint8_t *g2) {
    using namespace dw;
    const int NB = (int) fpk<ND>().fbytes;
    f3 Qx, Qy, u0, u1;
    D::f3_load_be(Qx, g2);
    D::f3_load_be(Qy, g2 + 3 * NB);
    D::f3_sqr(u0, Qx);
    fp_add<ND>(u0.c[0], u0.c[0], D::dk(c_d.ta));
    D::f3_mul(u0, u0, Qx);
    fp_add<ND>(u0.c[0], u0.c[0], D::dk(c_d.tb));
    D::f3_sqr(u1, Qy);
    const bool valid = D::f3_eq(u0, u1);
    D::f3_mul_fq(Qx, Qx, D::dk(c_d.nqrinv));
    D::f3_mul_fq(Qy, Qy, D::dk(c_d.nqrinv2));
    for (int i = 0; i < 3; i++) { put_fq(S_Qx0 + i, Qx.c[i]); put_fq(S_Qy0 + i, Qy.c[i]); }
    return valid;
  }

  // ---- the interpreter: the host flattened the pairing into a schedule (dw_sched.h: which level of which program runs when
  // depends on the curve's constants only); entry k + 1 is fetched while entry k runs ----
  static PBC_DEV void bzero_test() {
    using namespace dw;
    if (threadIdx.x == 0) {
      // B = 0 (the value after the easy part is +-1) must not poison 1 / (D B): invert D * 1 instead (d_final_exp)
      fq one, zero;
      fp_set<ND>(one, fpk<ND>().one);
#pragma unroll
      for (int k = 0; k < ND; k++) zero.v[k] = 0;
      fq b[3];
      bool b0 = true;
      for (int i = 0; i < 3; i++) { b[i] = get_fq(S_wB0 + i); b0 &= fp_is0<ND>(b[i]); }
      for (int i = 0; i < 3; i++) put_fq(S_Bn0 + i, b0 ? (i ? zero : one) : b[i]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  static PBC_DEV void inversion() {
    using namespace dw;
    if (threadIdx.x == 0) {
      fq n;
      fp_inv<ND>(n, get_fq(S_nrm0));                  // the only inversion
      put_fq(S_ninv, n);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // (an entry is the same for every lane: readfirstlane tells the compiler, so that the interpreter's branches are scalar
  // branches and not EXEC-masked regions -- where this compiler's SGPR spills into VGPR lanes go wrong, tools/gpu_faults.md)
  static PBC_DEV uint64_t uniform64(uint64_t v) {
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) v), hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (v >> 32));
    return ((uint64_t) hi << 32) | lo;
  }
  // A line of a pairing_pp table ([steps][3][ND] words, pairing_d.cuh d_pp_init_lane): lanes 0-2 fetch a', b', c' when a level
  // starts and put them into coefficient bank (line % 2) when it is over -- the fetch hides under the level's multiplications.
  struct Line { uint32_t w[ND]; };
  static PBC_DEV Line line_fetch(const uint32_t *tab, int line) {
    Line r;
    const uint32_t *p = tab + ((size_t) line * 3 + (threadIdx.x < 3 ? threadIdx.x : 0)) * ND;
#pragma unroll
    for (int k = 0; k < ND; k++) r.w[k] = p[k];
    return r;
  }
  static PBC_DEV void line_put(const Line &ln, int line) {
    if (threadIdx.x < 3) {
      fq v;
#pragma unroll
      for (int k = 0; k < ND; k++) v.v[k] = ln.w[k];
      put_fq(TB::line_slot(line) + (int) threadIdx.x, v);
    }
    __builtin_amdgcn_wave_barrier();
  }
  static PBC_DEV void run_entry(uint64_t e, const uint32_t *tab) {        // one LEVEL entry
    const int T = (int) ((e >> 34) & 15u);
    const Lev a = ent_a(e), b = ent_b(e);
    const Rows R = T > 4 && (PBC_DW_SPLIT || kSqueeze) ? load_rows4(a, b) : load_rows(a, b);
    if ((e >> 55) & 1u) {
      const int line = (int) ((e >> 42) & 4095u);
      const Line ln = line_fetch(tab, line);
      exec(R, R, T);
      line_put(ln, line);
    } else {
      exec(R, R, T);
    }
  }
  static __device__ __noinline__ void interpret(const uint64_t *sched_, const uint32_t *tab_ = nullptr) {
    const uint64_t *sched = reinterpret_cast<const uint64_t *>(uniform64(reinterpret_cast<uint64_t>(sched_)));
    const uint32_t *tab = reinterpret_cast<const uint32_t *>(uniform64(reinterpret_cast<uint64_t>(tab_)));
    uint64_t e = uniform64(sched[0]);
    for (int k = 1;; k++) {
      const int op = (int) ((e >> 38) & 15u);
#if PBC_DW_WHATIF & 1
      if (op != dw::OP_END) { e = uniform64(sched[k]); continue; }
#endif
      if (op == dw::OP_END) break;
      const uint64_t nxt = uniform64(sched[k]);       // (a scalar load that completes under this entry's work)
      if (op == dw::OP_LEVEL) run_entry(e, tab);
      else if (op == dw::OP_BZERO) bzero_test();
      else if (op == dw::OP_INV) inversion();
      else { const int line = (int) ((e >> 42) & 4095u); line_put(line_fetch(tab, line), line); }
      e = nxt;
    }
  }

  static PBC_DEV void begin() {
    const uint32_t *src = TB::rows_src();
    for (int i = (int) threadIdx.x; i < TB::kRows * 5; i += 64) g_lds_wv<ND, TB>[TB::kSlots * L + i] = src[i];
    for (int i = (int) threadIdx.x; i < TB::kSlots * L; i += 64) g_lds_wv<ND, TB>[i] = 0;
    __builtin_amdgcn_wave_barrier();
  }
  static PBC_DEV void store_gt(uint8_t *gt, bool valid) {
    using namespace dw;
    if (threadIdx.x < 6) {
      fq o = get_fq(S_f_x0 + (int) threadIdx.x);      // f.x0..2, f.y0..2 are consecutive slots: GT's wire order
      if (!valid) {                                   // an input that deserialises to O: the identity of GT
        fq one;
        fp_set<ND>(one, fpk<ND>().one);
#pragma unroll
        for (int k = 0; k < ND; k++) o.v[k] = threadIdx.x == 0 ? one.v[k] : 0u;
      }
      fp_store_be<ND>(gt + (size_t) threadIdx.x * fpk<ND>().fbytes, o);
    }
  }
  // element_pairing
  static __device__ void pairing(uint8_t *gt, const uint8_t *g1, const uint8_t *g2, const uint64_t *sched) {
    begin();
    __shared__ int valid_s;
    if (threadIdx.x == 0) { setup_consts(); const bool a = setup_point(g1), b = setup_twist(g2); valid_s = a && b ? 1 : 0; }
    __builtin_amdgcn_wave_barrier();
    interpret(sched);
    store_gt(gt, valid_s != 0);
  }
  // element_prod_pairing, first kernel: the Miller value of ONE TERM -> its record of the workspace (kRec words: the six slots
  // of f as they are, then the validity flag).  cc_millers_no_denom_affine (d_param.c:591-708) squares one accumulator for all
  // terms; the product of the terms' own Miller values is the same element of F_q^6 up to the lines' factors in F_q^*.
  static constexpr int kRec = (6 * L + 1 + 7) & ~7;   // 40 words for six limbs, 48 for seven, 56 for eight
  static __device__ void miller_term(uint32_t *rec, const uint8_t *g1, const uint8_t *g2, const uint64_t *sched) {
    begin();
    __shared__ int valid_s;
    if (threadIdx.x == 0) { setup_consts(); const bool a = setup_point(g1), b = setup_twist(g2); valid_s = a && b ? 1 : 0; }
    __builtin_amdgcn_wave_barrier();
    interpret(sched);
    if (threadIdx.x < 6 * L) rec[threadIdx.x] = slot(dw::S_f_x0)[threadIdx.x];
    if (threadIdx.x == 0) rec[6 * L] = (uint32_t) valid_s;
  }
  // second kernel: f <- the product of the k records (the two levels of f_mul0 on value bank 0, sched[0..1]), then the final
  // exponentiation (sched + 2); any invalid term: the identity
  static __device__ void finish(uint8_t *gt, const uint32_t *recs, int k, const uint64_t *sched_) {
    const uint64_t *sched = reinterpret_cast<const uint64_t *>(uniform64(reinterpret_cast<uint64_t>(sched_)));
    begin();
    if (threadIdx.x == 0) setup_consts();
    __builtin_amdgcn_wave_barrier();
    const int lane = (int) threadIdx.x, w = lane < 6 * L ? lane : 0;
    bool valid = recs[6 * L] != 0;
    if (lane < 6 * L) slot(dw::S_f_x0)[w] = recs[w];
    const uint64_t e0 = uniform64(sched[0]), e1 = uniform64(sched[1]);
    uint32_t nxt = k > 1 ? recs[kRec + w] : 0u;
    for (int t = 1; t < k; t++) {
      valid &= recs[(size_t) t * kRec + 6 * L] != 0;
      if (lane < 6 * L) slot(dw::S_V0_x0)[w] = nxt;
      __builtin_amdgcn_wave_barrier();
      if (t + 1 < k) nxt = recs[(size_t) (t + 1) * kRec + w];          // (in flight during the two levels)
      run_entry(e0, nullptr);
      run_entry(e1, nullptr);
    }
    interpret(sched + 2);
    store_gt(gt, valid);
  }
  // pairing_pp_apply: the lines' coefficients come from the table of pairing_pp_init (d_pp_init_lane), no arithmetic on E(F_q)
  static __device__ void pp_apply(uint8_t *gt, const uint32_t *tab, bool p_valid, const uint8_t *g2, const uint64_t *sched) {
    begin();
    __shared__ int valid_s;
    if (threadIdx.x == 0) { setup_consts(); valid_s = setup_twist(g2) ? 1 : 0; }
    __builtin_amdgcn_wave_barrier();
    interpret(sched, tab);
    store_gt(gt, p_valid && valid_s != 0);
  }
};

}  // namespace pbc
