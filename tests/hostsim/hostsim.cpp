// hostsim.cpp -- TEST INFRASTRUCTURE: the device headers compiled for the host, exported as a
// tiny C API (tests/test_hostsim.py).  It runs one lane at a time: same source, same limb
// arithmetic, same control flow as the HIP kernels.  Not part of the product.
#define PBC_HOSTSIM 1
#include <math.h>
#include <vector>
#include "../../pbc_amd/csrc/host_params.h"
#include "../../pbc_amd/csrc/pairing_al.cuh"
#include "../../pbc_amd/csrc/pairing_aw.cuh"
#include "../../pbc_amd/csrc/pairing_ew.cuh"
#include "../../pbc_amd/csrc/group_al.cuh"
#include "../../pbc_amd/csrc/group_l5.cuh"
#include "../../pbc_amd/csrc/group_more.cuh"

// run EXPR with N = the compile-time word count matching P->nlimb
#define HS_DISPATCH(nl, ...)                          \
  switch (nl) {                                       \
    case 5: { constexpr int N = 5; __VA_ARGS__; } break;     \
    case 6: { constexpr int N = 6; __VA_ARGS__; } break;     \
    case 7: { constexpr int N = 7; __VA_ARGS__; } break;     \
    case 8: { constexpr int N = 8; __VA_ARGS__; } break;     \
    case 16: { constexpr int N = 16; __VA_ARGS__; } break;   \
    case 33: { constexpr int N = 33; __VA_ARGS__; } break;   \
  }
// type f: 5- or 8-word fields
#define HS_DISPATCH_F(nl, ...)                         \
  switch (nl) {                                        \
    case 5: { constexpr int N = 5; __VA_ARGS__; } break;      \
    case 8: { constexpr int N = 8; __VA_ARGS__; } break;      \
  }
// types d / g: N words, degree DEG
#define HS_DISPATCH_D(P_, ...)                                                \
  switch ((P_)->nlimb * 8 + (P_)->deg) {                                      \
    case 5 * 8 + 3: { constexpr int N = 5, DEG = 3; __VA_ARGS__; } break;     \
    case 6 * 8 + 3: { constexpr int N = 6, DEG = 3; __VA_ARGS__; } break;     \
    case 7 * 8 + 3: { constexpr int N = 7, DEG = 3; __VA_ARGS__; } break;     \
    case 5 * 8 + 5: { constexpr int N = 5, DEG = 5; __VA_ARGS__; } break;     \
  }
// the object's constants into the block the kernel source reads (what a launch passes as KArgs<N> on the GPU)
static void activate(pbc_hip_pairing_s *P, bool for_pairing = false) {
  HS_DISPATCH(P->nlimb, { KArgs<N> K; fill_kargs<N>(P, K, for_pairing); memcpy(hostsim_kargs + sizeof hostsim_kargs - sizeof K, &K, sizeof K); });
}

// type a1 / type a outside the fast path on the same routines (AG<N>: the host mirror of its additive layer carries AL's bound
// tracker with the object's own limb width, filled limbs and subtraction constants): the two facts about q the bounds rest on
static bool ag_object(const pbc_hip_pairing_s *P) { return (P->type == '1' || (P->type == 'a' && P->a_generic)) && !P->ag_aux.empty(); }
template <int N>
static void ag_bind(const pbc_hip_pairing_s *P) {
  const int W = Limbs29<N>::W, L = Limbs29<N>::L, leff = (int) P->ag_aux[0];
  int qbits = 0;
  const uint32_t *q = N == 16 ? P->k16.p : P->k33.p;
  for (int i = 0; i < 32 * N; i++) if ((q[i >> 5] >> (i & 31)) & 1) qbits = i + 1;
  AG<N>::hs_tab = P->ag_aux.data();
  AG<N>::hs_leff = leff;
  AG<N>::hs_topf = ldexp(1.0, qbits - W * (leff - 1) - 1);
  AG<N>::hs_slackf = ldexp(1.0, W * L - qbits);
}
#define HS_DISPATCH_AG(P_, ...) do { if ((P_)->nlimb == 16) { constexpr int N = 16; ag_bind<N>(P_); __VA_ARGS__; } else { constexpr int N = 33; ag_bind<N>(P_); __VA_ARGS__; } } while (0)

extern "C" {

void *hostsim_init(const char *param, size_t len) {
  std::string type;
  if (!pbc_host::param_lookup(param, len, "type", type)) return nullptr;
  pbc_hip_pairing_s *P = new pbc_hip_pairing_s();
  int rc = 1;
  if (type == "a") { P->type = 'a'; rc = init_type_a(P, param, len); }
  else if (type == "a1") { P->type = '1'; rc = init_type_a1(P, param, len); }
  else if (type == "e") { P->type = 'e'; rc = init_type_e(P, param, len); }
  else if (type == "d") { P->type = 'd'; rc = init_type_d(P, param, len, 3); }
  else if (type == "g") { P->type = 'g'; rc = init_type_d(P, param, len, 5); }
  else if (type == "f") { P->type = 'f'; rc = init_type_f(P, param, len); }
  if (rc) { delete P; return nullptr; }
  activate(P);
  // the device-side derivations of the library, stage by stage (each stage sees the previous one's constants)
  if (P->type == 'd' || P->type == 'g') {
    DConst tmp;
    HS_DISPATCH_D(P, TypeMNT<N, DEG>::init_stage1(&tmp, P->draw, c_d));
    P->dconst = tmp;
    activate(P);
    HS_DISPATCH_D(P, TypeMNT<N, DEG>::init_stage2(&tmp, P->draw));
    P->dconst = tmp;
  }
  if (P->type == 'e') {
    EConst tmp;
    if (P->nlimb == 16) e_init_lane<16>(&tmp, P->eraw, c_e);
    else e_init_lane<33>(&tmp, P->eraw, c_e);
    P->econst = tmp;
  }
  if (P->type == 'f') {
    FConst tmp;
    HS_DISPATCH_F(P->nlimb, TypeF<N>::init_stage1(&tmp, P->fraw, c_f));
    P->fconst = tmp;
    activate(P);
    HS_DISPATCH_F(P->nlimb, TypeF<N>::init_stage2(&tmp, P->fraw));
    P->fconst = tmp;
    if (P->fraw.e4bits > 0) {            // as ensure_derived: the i-basis copy for the pairing kernels
      activate(P);
      HS_DISPATCH_F(P->nlimb, TypeF<N>::init_stage3(&tmp, P->fraw));
      P->fconst_i = tmp;
      P->f_bm1 = tmp.bm1 != 0;
      if (P->f_bm1 && P->fraw.xs_try) {  // as derive_f: the sparse xi of the pairing kernels
        activate(P, true);
        HS_DISPATCH_F(P->nlimb, TypeF<N>::init_stage4(&tmp, P->fraw));
        P->fconst_i = tmp;
      }
    }
  }
  activate(P);
  return P;
}
const char *hostsim_error() { return g_err; }
// The subtraction constants of the limb-form type a kernel (AConst::ksub): rebuilt and checked by the routine init itself
// runs (hostbn.h ksub_build: sum_i k_i 2^(29 i) == c q, k_i >= D (2^29 - 1) below the top, k_i < 2^32, q >= 2^504);
// returns the number of violations, counting a stored constant that differs from the rebuilt one.
int hostsim_check_ksub(void *h) {
  using pbc_host::Big;
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  if (P->type != 'a' || P->a_generic) return -1;
  static const uint32_t cd[5][2] = {{2, 1}, {4, 2}, {8, 4}, {12, 2}, {16, 2}};
  Big q;
  q.w.assign(P->k16.p, P->k16.p + 16);
  q.trim();
  int bad = 0;
  for (int t = 0; t < 5; t++) {
    uint32_t k[18];
    bad += pbc_host::ksub_build(q, 18, 505, cd[t][0], cd[t][1], k);
    for (int i = 0; i < 18; i++) bad += k[i] != P->a.ksub[t][i];
  }
  return bad;
}
// multiply-adds executed by the kernel source since the last reset (PBC_COUNT_MACS hooks of fp.cuh)
uint64_t hostsim_macs_read(int reset) { uint64_t v = hostsim_macs; if (reset) hostsim_macs = 0; return v; }
int hostsim_lens(void *h, int *l1, int *l2, int *lt) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  *l1 = P->len1; *l2 = P->len2; *lt = P->lenT;
  return 0;
}
// element_pairing on the one-pairing-per-wavefront routine (pairing_aw.cuh), type a with the 512-bit field
int hostsim_pairing_wave(void *h, uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  if (P->type == 'e' && !P->ag_aux.empty()) {              // pairing_ew.cuh: type e on the same routines
    activate(P, true);
    HS_DISPATCH_AG(P, for (size_t u = 0; u < n; u++) { EW<N, 1> w; w.pairing_wave(gt + u * P->lenT, g1 + u * P->len1, g2 + u * P->len2); });
    activate(P);
    return 0;
  }
  if (ag_object(P)) {
    activate(P, true);
    // (NW = 8: the three-round schedule of the smallest batches, miller_loop_p; products below run the five-round one)
    HS_DISPATCH_AG(P, for (size_t u = 0; u < n; u++) { AW<N, 8, AG<N>> w; w.pairing_wave(gt + u * P->lenT, g1 + u * P->len1, g2 + u * P->len2); });
    activate(P);
    return 0;
  }
  if (P->type != 'a' || P->a_generic) return 1;
  activate(P, true);
  for (size_t u = 0; u < n; u++) { AW<16, 1> w; w.pairing_wave(gt + u * P->lenT, g1 + u * P->len1, g2 + u * P->len2); }
  activate(P);
  return 0;
}
// pairing_pp_apply and k-term products on the wave routines (pairing_aw.cuh pp_apply_wave / miller_record_wave /
// prod_finish_wave); the table is the one the library's a_pp_init_lane writes
int hostsim_pp_wave(void *h, uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  if (ag_object(P)) {                    // the table a1_pp_init_lane writes: one entry per doubling and per addition
    activate(P, true);
    int steps = P->a.rbits - 1;
    for (int m = 1; m <= P->a.rbits - 2; m++) steps += ((P->a.r[m >> 5] | P->a.rm[m >> 5]) >> (m & 31)) & 1;
    HS_DISPATCH_AG(P, {
      std::vector<uint32_t> tab((size_t) steps * 3 * N);
      const bool ok = a1_pp_init_lane<N>(tab.data(), g1);
      for (size_t u = 0; u < n; u++) { AW<N, 1, AG<N>> w; w.pp_apply_wave(gt + u * P->lenT, tab.data(), ok, g2 + u * P->len2); }
    });
    activate(P);
    return 0;
  }
  if (P->type != 'a' || P->a_generic) return 1;
  activate(P, true);
  std::vector<uint32_t> tab((size_t) (P->a.exp2 + 1) * 3 * 16);
  const bool ok = a_pp_init_lane<16>(tab.data(), g1);
  for (size_t u = 0; u < n; u++) { AW<16, 1> w; w.pp_apply_wave(gt + u * P->lenT, tab.data(), ok, g2 + u * P->len2); }
  activate(P);
  return 0;
}
int hostsim_prod_wave(void *h, uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n, int k) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  if (P->type == 'e' && !P->ag_aux.empty()) {
    activate(P, true);
    HS_DISPATCH_AG(P, for (size_t u = 0; u < n; u++) {
      typedef EW<N, 1> WV;
      std::vector<typename WV::wrec> rec((size_t) k);
      for (int t = 0; t < k; t++) { WV w; w.miller_record_wave(rec[t], g1 + (u * k + t) * P->len1, g2 + (u * k + t) * P->len2); }
      WV w;
      w.prod_finish_wave(gt + u * P->lenT, rec.data(), k);
    });
    activate(P);
    return 0;
  }
  if (ag_object(P)) {
    activate(P, true);
    HS_DISPATCH_AG(P, for (size_t u = 0; u < n; u++) {
      typedef AW<N, 1, AG<N>> WV;
      std::vector<typename WV::wrec> rec((size_t) k);
      for (int t = 0; t < k; t++) { WV w; w.miller_record_wave(rec[t], g1 + (u * k + t) * P->len1, g2 + (u * k + t) * P->len2); }
      WV w;
      w.prod_finish_wave(gt + u * P->lenT, rec.data(), k);
    });
    activate(P);
    return 0;
  }
  if (P->type != 'a' || P->a_generic) return 1;
  activate(P, true);
  for (size_t u = 0; u < n; u++) {
    std::vector<AW<16, 1>::wrec> rec((size_t) k);
    for (int t = 0; t < k; t++) { AW<16, 1> w; w.miller_record_wave(rec[t], g1 + (u * k + t) * P->len1, g2 + (u * k + t) * P->len2); }
    AW<16, 1> w;
    w.prod_finish_wave(gt + u * P->lenT, rec.data(), k);
  }
  activate(P);
  return 0;
}
// n units of k terms each, one lane after the other
int hostsim_prod_pairing(void *h, uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n, int k) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P, true);                     // the constant block of a pairing launch (launch_pairing / launch_prod)
  static uint32_t lds[72];               // Q of the register-resident fields (2 N words) / the two hot elements of the 33-word steps (72 words)
  for (size_t u = 0; u < n; u++) {
    const uint8_t *a = g1 + u * k * P->len1, *b = g2 + u * k * P->len2;
    uint8_t *o = gt + u * P->lenT;
    if (P->type == 'a' && !P->a_generic && k == 1) AL<16>::pairing_lane(o, a, b);     // as launch_pairing does
    else if (P->type == 'a' && !P->a_generic && !P->a_prod_shared) {                 // as launch_prod does: one term per lane, then the product
      std::vector<uint4> ws((size_t) k * AL<16>::MREC);
      for (int j = 0; j < k; j++) AL<16>::miller_record_lane(ws.data() + (size_t) j * AL<16>::MREC, a + (size_t) j * P->len1, b + (size_t) j * P->len2);
      AL<16>::prod_finish_lane(o, ws.data(), k);
    }
    else if (P->type == 'a' && !P->a_generic) { std::vector<uint4> ws((size_t) k * 24 * 128); a_prod_pairing_lane<16>(o, a, b, k, ws.data(), lds, 1); }
    else if ((P->type == 'a' || P->type == '1') && P->nlimb == 16) a1_prod_pairing_lane<16>(o, a, b, k, lds, 1);
    else if (P->type == '1' || P->type == 'a') a1_prod_pairing_lane<33>(o, a, b, k, lds, 1);
    else if (P->type == 'e' && P->nlimb == 16) e_prod_pairing_lane<16>(o, a, b, k, lds, 1);
    else if (P->type == 'e') e_prod_pairing_lane<33>(o, a, b, k, lds, 1);
    else if (P->type == 'd' || P->type == 'g') { HS_DISPATCH_D(P, { std::vector<uint32_t> ws((size_t) k * TypeMNT<N, DEG>::DL_WORDS * 128); TypeMNT<N, DEG>::d_prod_pairing_lane(o, a, b, k, ws.data()); }); }
    else if (P->f_bm1 && P->fconst_i.xs_ok && P->nlimb == 5) TypeF<5, true, true>::f_prod_pairing_lane(o, a, b, k);       // as launch_f
    else if (P->f_bm1) { HS_DISPATCH_F(P->nlimb, (TypeF<N, true>::f_prod_pairing_lane(o, a, b, k))); }
    else { HS_DISPATCH_F(P->nlimb, TypeF<N>::f_prod_pairing_lane(o, a, b, k)); }
  }
  return 0;
}
// pairing_pp_init + pairing_pp_apply over n second arguments (Type A)
int hostsim_pp(void *h, uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  static uint32_t tab[512 * 3 * 16];
  if (P->type == 'd' || P->type == 'g') {
    HS_DISPATCH_D(P, {
      bool v = TypeMNT<N, DEG>::d_pp_init_lane(tab, g1);
      for (size_t u = 0; u < n; u++) TypeMNT<N, DEG>::d_pp_apply_lane(gt + u * P->lenT, tab, v, g2 + u * P->len2);
    });
    return 0;
  }
  if (P->nlimb == 16 && ((P->type == 'a' && P->a_generic) || P->type == '1')) {
    static uint32_t tab2[1024 * 3 * 16];
    bool v = a1_pp_init_lane<16>(tab2, g1);
    for (size_t u = 0; u < n; u++) a1_pp_apply_lane<16>(gt + u * P->lenT, tab2, v, g2 + u * P->len2);
    return 0;
  }
  if (P->type == '1' || (P->type == 'a' && P->a_generic)) {
    static uint32_t tab1[2048 * 3 * 33];
    bool v = a1_pp_init_lane<33>(tab1, g1);
    static uint32_t hot[72];             // as the kernel: f^2 of a step in the lane's two LDS slots
    for (size_t u = 0; u < n; u++) a1_pp_apply_lane<33>(gt + u * P->lenT, tab1, v, g2 + u * P->len2, hot);
    return 0;
  }
  if (P->type != 'a') return 1;
  bool v = a_pp_init_lane<16>(tab, g1);
  for (size_t u = 0; u < n; u++) AL<16>::pp_apply_lane(gt + u * P->lenT, tab, v, g2 + u * P->len2);
  return 0;
}
// compressed / x-only points on G1: dir 0 compress, 1 decompress, 2 to x-only, 3 from x-only
int hostsim_compress(void *h, int dir, uint8_t *out, const uint8_t *in, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  if (!P->hash.ts_ready) {
    HS_DISPATCH(P->nlimb, fp_ts_init<N>(P->hash.ts_c, P->hash.ts_t, P->hash.ts_tbits, P->hash.half, P->hash.halfbits));
    P->hash.ts_ready = true;
    activate(P);
  }
  const size_t lp = P->len1, lc = P->len_fq + (dir < 2 ? 1 : 0);
  for (size_t i = 0; i < n; i++) {
    if (dir == 0) { HS_DISPATCH(P->nlimb, g_compress_lane<N>(out + i * lc, in + i * lp)); }
    else if (dir == 1) { HS_DISPATCH(P->nlimb, g_decompress_lane<N>(out + i * lp, in + i * lc)); }
    else if (dir == 2) { HS_DISPATCH(P->nlimb, g_to_x_only_lane<N>(out + i * lc, in + i * lp)); }
    else { HS_DISPATCH(P->nlimb, g_decompress_lane<N>(out + i * lp, in + i * lc, true)); }
  }
  return 0;
}
// twists (G2 of types d, g, f): what 0 element_from_hash (digests of hlen bytes), 1 compress, 2 decompress,
// 3 to x-only, 4 from x-only
#define HS_DISPATCH_TWIST(P_, ...)                                                  \
  do {                                                                              \
    if ((P_)->type == 'f') { HS_DISPATCH_F((P_)->nlimb, { typedef Fq2Ops<N> F; __VA_ARGS__; }); } \
    else { HS_DISPATCH_D(P_, { typedef FdOps<N, DEG> F; __VA_ARGS__; }); }          \
  } while (0)
int hostsim_g2_points(void *h, int what, uint8_t *out, const uint8_t *in, int hlen, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  if (P->type != 'd' && P->type != 'g' && P->type != 'f') return 1;
  activate(P);
  if (!P->xs_ready) {
    HS_DISPATCH_TWIST(P, ext_ts_init<F>(P->xs.c));
    P->xs_ready = true;
    activate(P);
  }
  if (P->type == 'f' && !P->hash.ts_ready) {             // fq_sqrt goes through square roots in F_q
    HS_DISPATCH(P->nlimb, fp_ts_init<N>(P->hash.ts_c, P->hash.ts_t, P->hash.ts_tbits, P->hash.half, P->hash.halfbits));
    P->hash.ts_ready = true;
    activate(P);
  }
  const size_t lp = P->len2, lc = lp / 2 + 1, lx = lp / 2;
  const size_t li = what == 0 ? (size_t) hlen : (what == 1 || what == 3) ? lp : what == 2 ? lc : lx;
  const size_t lo = what == 1 ? lc : what == 3 ? lx : lp;
  for (size_t i = 0; i < n; i++) {
    if (what == 0) { HS_DISPATCH_TWIST(P, g2_from_hash_lane<F>(out + i * lo, in + i * li, hlen)); }
    else if (what == 1) { HS_DISPATCH_TWIST(P, g2_compress_lane<F>(out + i * lo, in + i * li)); }
    else if (what == 2) { HS_DISPATCH_TWIST(P, g2_decompress_lane<F>(out + i * lo, in + i * li)); }
    else if (what == 3) { memcpy(out + i * lo, in + i * li, lx); }
    else { HS_DISPATCH_TWIST(P, g2_from_x_lane<F>(out + i * lo, in + i * li)); }
  }
  return 0;
}
// element_mul_zn on G2 of the asymmetric types (twists)
static int hostsim_slow_group = 0;     // 1: only the complete word-form routines (what the library runs for reported lanes)
static uint64_t hostsim_fallbacks = 0; // lanes the fast routines reported
int hostsim_g2_mul(void *h, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  for (size_t i = 0; i < n; i++) {
    if (!hostsim_slow_group && (P->type == 'd' || P->type == 'g' || P->type == 'f')) {   // as the library: the windowed ladder first
      bool ok = false;
      if (P->type == 'f') { HS_DISPATCH_F(P->nlimb, ok = (ec_mul_win_lane<Fq2Ops<N>>(out + i * P->len2, a + i * P->len2, b + i * P->len_zr, P->len_zr))); }
      else { HS_DISPATCH_D(P, ok = (ec_mul_win_lane<FdOps<N, DEG>>(out + i * P->len2, a + i * P->len2, b + i * P->len_zr, P->len_zr))); }
      if (ok) continue;
      hostsim_fallbacks++;
    }
    if (P->type == 'd' || P->type == 'g') {
      HS_DISPATCH_D(P, (ec_mul_lane<FdOps<N, DEG>>(out + i * P->len2, a + i * P->len2, b + i * P->len_zr, P->len_zr)));
    } else if (P->type == 'f') {
      HS_DISPATCH_F(P->nlimb, (ec_mul_lane<Fq2Ops<N>>(out + i * P->len2, a + i * P->len2, b + i * P->len_zr, P->len_zr)));
    } else {
      HS_DISPATCH(P->nlimb, g_mul_lane<N>(out + i * P->len2, a + i * P->len2, b + i * P->len_zr, P->len_zr));
    }
  }
  return 0;
}
// fixed-base powers (element_pp_init + element_pp_pow_zn): group 1 / 2 / 3; the table is built entry by entry as the
// library's kernels do, then every scalar takes the table routine (and the complete ladder when that reports the lane)
int hostsim_element_pp(void *h, int group, uint8_t *out, const uint8_t *base, const uint8_t *zr, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  const int zlen = P->len_zr;
  const bool twist = group == 2 && (P->type == 'd' || P->type == 'g' || P->type == 'f');
  if (group == 3) {
    const size_t units = (size_t) zlen << kPpWin;
#define HS_GT_PP(G_)                                                                           \
    { std::vector<uint32_t> tab(units * G_::WORDS_EL);                                          \
      for (size_t u = 0; u < units; u++) gt_pp_entry_lane<G_>(tab.data(), base, u);             \
      for (size_t i = 0; i < n; i++) gt_pp_pow_lane<G_>(out + i * P->lenT, tab.data(), zr + i * zlen, zlen); }
    if (P->type == 'a' || P->type == '1') { if (P->nlimb == 16) HS_GT_PP(GtA<16>) else HS_GT_PP(GtA<33>) }
    else if (P->type == 'e') { if (P->nlimb == 16) HS_GT_PP(GtE<16>) else HS_GT_PP(GtE<33>) }
    else if (P->type == 'f') { HS_DISPATCH_F(P->nlimb, HS_GT_PP(GtF<N>)); }
    else { HS_DISPATCH_D(P, { typedef GtD<N, DEG> GD; HS_GT_PP(GD) }); }
#undef HS_GT_PP
    return 0;
  }
  const size_t units = (size_t) zlen * kPpRowLen;
  const size_t lp = group == 2 ? P->len2 : P->len1;
  // as the library: G1 of the 5-word fields takes the limb-form table routine (group_l5.cuh)
  const bool limb5 = group == 1 && P->nlimb == 5 && ((P->type == 'd' && P->deg == 3 && P->dconst.limb_ok) || (P->type == 'f' && P->fconst.pl_ok));
#define HS_EC_PP(F_)                                                                           \
  { std::vector<uint32_t> tab(units * 2 * F_::WORDS_EL);                                        \
    std::vector<uint8_t> flags(units);                                                          \
    for (size_t u = 0; u < units; u++) ec_pp_entry_lane<F_>(tab.data(), flags.data(), base, zlen, u); \
    bool complete_only = false;                                                                 \
    for (uint8_t f : flags) complete_only |= f != 0;                                            \
    for (size_t i = 0; i < n; i++) {                                                            \
      if (!complete_only && limb5 && (P->type == 'd' ? GL<5, KPd>::pp_pow_lane(out + i * lp, tab.data(), zr + i * zlen, zlen)  \
                                                      : GL<5, KPf>::pp_pow_lane(out + i * lp, tab.data(), zr + i * zlen, zlen))) continue; \
      if (!complete_only && !limb5 && ec_pp_pow_lane<F_>(out + i * lp, tab.data(), zr + i * zlen, zlen)) continue; \
      hostsim_fallbacks++;                                                                      \
      ec_mul_lane<F_>(out + i * lp, base, zr + i * zlen, zlen);                                 \
    } }
  if (twist && P->type == 'f') { HS_DISPATCH_F(P->nlimb, HS_EC_PP(Fq2Ops<N>)); }
  else if (twist) { HS_DISPATCH_D(P, { typedef FdOps<N, DEG> FD; HS_EC_PP(FD) }); }
  else { HS_DISPATCH(P->nlimb, HS_EC_PP(FqOps<N>)); }
#undef HS_EC_PP
  return 0;
}
// group operations: what 0 = G mul_zn, 1 = GT mul, 2 = GT pow
void hostsim_group_mode(int slow) { hostsim_slow_group = slow; }
uint64_t hostsim_group_fallbacks(int reset) { uint64_t v = hostsim_fallbacks; if (reset) hostsim_fallbacks = 0; return v; }
int hostsim_group(void *h, int what, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  for (size_t i = 0; i < n; i++) {
    if (what == 0) {
      // as the library: the limb-form ladder first where it exists, the complete routine for the lanes it reports
      if (P->type == 'a' && !P->a_generic && !hostsim_slow_group) {
        if (GAL<16>::gmul_lane(out + i * P->len1, a + i * P->len1, b + i * P->len_zr, P->len_zr)) continue;
        hostsim_fallbacks++;
      } else if (!hostsim_slow_group && P->nlimb == 5 && ((P->type == 'd' && P->deg == 3 && P->dconst.limb_ok) || (P->type == 'f' && P->fconst.pl_ok))) {
        const bool ok = P->type == 'd' ? GL<5, KPd>::gmul_lane(out + i * P->len1, a + i * P->len1, b + i * P->len_zr, P->len_zr)
                                       : GL<5, KPf>::gmul_lane(out + i * P->len1, a + i * P->len1, b + i * P->len_zr, P->len_zr);
        if (ok) continue;
        hostsim_fallbacks++;
      } else if (!hostsim_slow_group) {
        bool ok = false;
        HS_DISPATCH(P->nlimb, ok = ec_mul_win_lane<FqOps<N>>(out + i * P->len1, a + i * P->len1, b + i * P->len_zr, P->len_zr));
        if (ok) continue;
        hostsim_fallbacks++;
      }
      HS_DISPATCH(P->nlimb, g_mul_lane<N>(out + i * P->len1, a + i * P->len1, b + i * P->len_zr, P->len_zr));
    } else {
      uint8_t *o = out + i * P->lenT;
      const uint8_t *x = a + i * P->lenT, *y = b + i * (what == 1 ? P->lenT : P->len_zr);
      if (what == 3) {                 // pairing->finalpow
        if (P->type == 'a' || P->type == '1') { if (P->nlimb == 16) a_finalpow_lane<16>(o, x); else a_finalpow_lane<33>(o, x); }
        else if (P->type == 'e') { HS_DISPATCH(P->nlimb, e_finalpow_lane<N>(o, x)); }
        else if (P->type == 'd' || P->type == 'g') { HS_DISPATCH_D(P, (d_finalpow_lane<N, DEG>(o, x))); }
        else { HS_DISPATCH_F(P->nlimb, f_finalpow_lane<N>(o, x)); }
        continue;
      }
      if (what == 2 && P->type == 'a' && !P->a_generic && !hostsim_slow_group) {
        if (GAL<16>::gt_pow_lane(o, x, y, P->len_zr)) continue;
        hostsim_fallbacks++;
      }
      if ((P->type == 'a' || P->type == '1') && P->nlimb == 16) { if (what == 1) a_gt_mul_lane<16>(o, x, y); else a_gt_pow_lane<16>(o, x, y, P->len_zr); }
      else if (P->type == '1' || P->type == 'a') { if (what == 1) a_gt_mul_lane<33>(o, x, y); else a_gt_pow_lane<33>(o, x, y, P->len_zr); }
      else if (P->type == 'e') {
        HS_DISPATCH(P->nlimb, {
          fp<N> u, v;
          fp_load_be<N>(u, x);
          if (what == 1) { fp_load_be<N>(v, y); fp_mul<N>(u, u, v); }
          else {
            fp<N> acc, t;
            fp_set<N>(acc, fpk<N>().one);
            for (int bi = 8 * P->len_zr - 1; bi >= 0; bi--) { fp_sqr<N>(acc, acc); fp_mul<N>(t, acc, u); fp_cmov<N>(acc, t, zr_bit(y, P->len_zr, bi) != 0); }
            u = acc;
          }
          fp_store_be<N>(o, u);
        });
      }
      else if (P->type == 'd' || P->type == 'g') { HS_DISPATCH_D(P, if (what == 1) d_gt_mul_lane<N, DEG>(o, x, y); else d_gt_pow_lane<N, DEG>(o, x, y, P->len_zr)); }
      else {
        if (what == 2 && P->nlimb == 5 && !hostsim_slow_group) {     // as the library: cyclotomic squarings in the pairing kernels' basis first
          activate(P, true);
          bool ok;
          if (P->f_bm1 && P->fconst_i.xs_ok) ok = f_gt_pow_cyc_lane<TypeF<5, true, true>>(o, x, y, P->len_zr);
          else if (P->f_bm1) ok = f_gt_pow_cyc_lane<TypeF<5, true, false>>(o, x, y, P->len_zr);
          else ok = f_gt_pow_cyc_lane<TypeF<5, false, false>>(o, x, y, P->len_zr);
          activate(P);
          if (ok) continue;
          hostsim_fallbacks++;
        }
        HS_DISPATCH_F(P->nlimb, if (what == 1) f_gt_mul_lane<N>(o, x, y); else f_gt_pow_lane<N>(o, x, y, P->len_zr));
      }
    }
  }
  return 0;
}
// element_from_hash on G1 (Type A): n digests of hlen bytes
int hostsim_from_hash(void *h, uint8_t *out, const uint8_t *data, int hlen, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  if (!P->hash.ts_ready) {
    HS_DISPATCH(P->nlimb, fp_ts_init<N>(P->hash.ts_c, P->hash.ts_t, P->hash.ts_tbits, P->hash.half, P->hash.halfbits));
    P->hash.ts_ready = true;
    activate(P);
  }
  for (size_t i = 0; i < n; i++) {
    if (P->type == 'a' && !P->a_generic && !hostsim_slow_group) {       // as the library: the limb-form routine first
      if (GAL<16>::from_hash_lane(out + i * P->len1, data + i * hlen, hlen)) continue;
      hostsim_fallbacks++;
    }
    HS_DISPATCH(P->nlimb, g_from_hash_lane<N>(out + i * P->len1, data + i * hlen, hlen));
  }
  return 0;
}
// diagnostics mirroring pbc_hip_diag_stage
int hostsim_stage(void *h, int stage, uint8_t *out, size_t out_len, const uint8_t *g1, const uint8_t *g2, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  if (stage == 0) {
    const void *src = (P->type == 'd' || P->type == 'g') ? (const void *) &P->dconst : P->type == 'f' ? (const void *) &P->fconst : (const void *) &P->a;
    size_t len = (P->type == 'd' || P->type == 'g') ? sizeof P->dconst : P->type == 'f' ? sizeof P->fconst : sizeof P->a;
    memcpy(out, src, len < out_len ? len : out_len);
    return (int) len;
  }
  if (stage == 1 && P->type == 'f') {
    for (size_t u = 0; u < n; u++) { HS_DISPATCH_F(P->nlimb, TypeF<N>::f_prod_pairing_lane(out + u * P->lenT, g1 + u * P->len1, g2 + u * P->len2, 1, true)); }
    return 0;
  }
  if (stage >= 10 && P->type == 'f') {
    for (size_t u = 0; u < n; u++) { HS_DISPATCH_F(P->nlimb, TypeF<N>::f_debug_lane(stage, out + u * P->lenT, g1 + u * P->lenT, g2 + u * P->lenT)); }
    return 0;
  }
  return -1;
}
// F_q ops on canonical bytes (same switch as fq_op_kernel)
int hostsim_fq_op(void *h, int op, uint8_t *c, const uint8_t *a, const uint8_t *b, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  const int L = P->len_fq;
  for (size_t i = 0; i < n; i++) {
    HS_DISPATCH(P->nlimb, {
      fp<N> x, y, z;
      fp_load_be<N>(x, a + i * L); fp_load_be<N>(y, b + i * L);
      switch (op) { case 0: fp_mul<N>(z, x, y); break; case 1: fp_add<N>(z, x, y); break; case 2: fp_sub<N>(z, x, y); break;
        case 3: fp_inv<N>(z, x); break; case 4: fp_neg<N>(z, x); break; case 5: fp_halve<N>(z, x); break; default: fp_dbl<N>(z, x); }
      fp_store_be<N>(c + i * L, z);
    });
  }
  return 0;
}

// round 5 (group_more.cuh): the group law on G1 / G2, the multi-exponentiations on G1 / G2 / GT, Z_r arithmetic --
// the lane bodies the kernels of pbc_hip_group2.hip run, one unit per call
#define HS_DISPATCH_G(P_, group_, ...)                                                                 \
  do {                                                                                                 \
    const bool sym_ = (P_)->type == 'a' || (P_)->type == '1' || (P_)->type == 'e';                      \
    if ((group_) == 2 && !sym_) { HS_DISPATCH_TWIST(P_, __VA_ARGS__); }                                \
    else { HS_DISPATCH((P_)->nlimb, { typedef FqOps<N> F; __VA_ARGS__; }); }                           \
  } while (0)
int hostsim_affine_op(void *h, int op, int group, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  const size_t L = (size_t) (group == 2 ? P->len2 : P->len1);
  for (size_t i = 0; i < n; i++) HS_DISPATCH_G(P, group, ec_affine_op_lane<F>(op, out + i * L, a + i * L, b ? b + i * L : a + i * L));
  return 0;
}
int hostsim_multi(void *h, int group, int k, uint8_t *out, const uint8_t *a1, const uint8_t *n1, const uint8_t *a2, const uint8_t *n2,
                  const uint8_t *a3, const uint8_t *n3, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  const size_t L = (size_t) (group == 1 ? P->len1 : group == 2 ? P->len2 : P->lenT);
  MultiArgs M;
  M.a[0] = a1; M.a[1] = a2; M.a[2] = k > 2 ? a3 : a1;
  M.z[0] = n1; M.z[1] = n2; M.z[2] = k > 2 ? n3 : n1;
  M.astride = L;
  M.zstride = (size_t) P->len_zr;
  for (size_t i = 0; i < n; i++) {
    if (group == 3) {
      if (P->type == 'a' || P->type == '1') { if (P->nlimb == 16) gt_multi_pow_lane<GtA<16>>(out + i * L, M, i, k, P->len_zr); else gt_multi_pow_lane<GtA<33>>(out + i * L, M, i, k, P->len_zr); }
      else if (P->type == 'e') { if (P->nlimb == 16) gt_multi_pow_lane<GtE<16>>(out + i * L, M, i, k, P->len_zr); else gt_multi_pow_lane<GtE<33>>(out + i * L, M, i, k, P->len_zr); }
      else if (P->type == 'f') { HS_DISPATCH_F(P->nlimb, gt_multi_pow_lane<GtF<N>>(out + i * L, M, i, k, P->len_zr)); }
      else { HS_DISPATCH_D(P, (gt_multi_pow_lane<GtD<N, DEG>>(out + i * L, M, i, k, P->len_zr))); }
    } else {
      // as the library: the fast pass, and the complete routine for the lanes it reports
      bool ok = false;
      if (!hostsim_slow_group && P->type == 'a' && !P->a_generic && (k == 2 || k == 3)) {
        // type a on the 512-bit field: the joint limb-form ladder (group_al.cuh gmulk_lane), bound tracker armed
        const uint8_t *pa[3], *pz[3];
        for (int j = 0; j < 3; j++) { pa[j] = M.a[j] + i * M.astride; pz[j] = M.z[j] + i * M.zstride; }
        ok = k == 2 ? GAL<16>::gmulk_lane<2>(out + i * L, pa, pz, P->len_zr) : GAL<16>::gmulk_lane<3>(out + i * L, pa, pz, P->len_zr);
      } else if (!hostsim_slow_group && group == 1 && (k == 2 || k == 3) && P->nlimb == 5 &&
                 ((P->type == 'd' && P->deg == 3 && P->dconst.limb_ok) || (P->type == 'f' && P->fconst.pl_ok))) {
        // G1 of the five-word fields: the same on group_l5.cuh
        const uint8_t *pa[3], *pz[3];
        for (int j = 0; j < 3; j++) { pa[j] = M.a[j] + i * M.astride; pz[j] = M.z[j] + i * M.zstride; }
        if (P->type == 'd') ok = k == 2 ? GL<5, KPd>::gmulk_lane<2>(out + i * L, pa, pz, P->len_zr) : GL<5, KPd>::gmulk_lane<3>(out + i * L, pa, pz, P->len_zr);
        else ok = k == 2 ? GL<5, KPf>::gmulk_lane<2>(out + i * L, pa, pz, P->len_zr) : GL<5, KPf>::gmulk_lane<3>(out + i * L, pa, pz, P->len_zr);
      } else
      if (!hostsim_slow_group) { HS_DISPATCH_G(P, group, ok = ec_multi_mul_fast_lane<F>(out + i * L, M, i, k, P->len_zr)); }
      if (!ok) {
        if (!hostsim_slow_group) hostsim_fallbacks++;
        HS_DISPATCH_G(P, group, ec_multi_mul_lane<F>(out + i * L, M, i, k, P->len_zr));
      }
    }
  }
  return 0;
}
int hostsim_zr_op(void *h, int op, uint8_t *out, const uint8_t *a, const uint8_t *b, int hlen, size_t n) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  if (!P->zr_nlimb) return 1;
  int rc = 0;
  HS_DISPATCH(P->zr_nlimb, {
    KArgs<N> K;
    rc = zr_kargs<N>(P, K);
    memcpy(hostsim_kargs + sizeof hostsim_kargs - sizeof K, &K, sizeof K);
    const size_t L = (size_t) P->len_zr;
    for (size_t i = 0; i < n && !rc; i++) zr_op_lane<N>(op, out + i * L, a + i * (op == 8 ? (size_t) hlen : L), b ? b + i * L : nullptr, hlen);
  });
  activate(P);
  return rc;
}
int hostsim_len_zr(void *h) { return ((pbc_hip_pairing_s *) h)->len_zr; }
// one fused product of fp.cuh ("Fused products": fp_mulx / fp_sqrx) on residues given as little-endian words, and the type e
// constant it is parameterised by: ops = a, a2, b, b2, c1, c2, d2 (N words each; sqr = 2: fp_sopx computes a b +- a2 (k b2 +- d2) ...); out = r
int hostsim_fx(void *h, int sqr, int op, const uint32_t *ops, uint32_t *out) {
  pbc_hip_pairing_s *P = (pbc_hip_pairing_s *) h;
  activate(P);
  HS_DISPATCH(P->nlimb, {
    fp<N> e[7], r;
    for (int j = 0; j < 7; j++) fp_set<N>(e[j], ops + j * N);
    if (sqr == 2) fp_sopx<N>(r, op, e[0], e[2], e[1], e[3], e[6], e[4], e[5]);       // a b +- a2 (k b2 +- d2) ...
    else if (sqr) fp_sqrx<N>(r, op, e[0], e[1], e[4], e[5]);
    else fp_mulx<N>(r, op, e[0], e[1], e[2], e[3], e[4], e[5]);
    for (int k = 0; k < N; k++) out[k] = r.v[k];
  });
  return P->nlimb;
}
int hostsim_e_rxs(void *h) { return ((pbc_hip_pairing_s *) h)->econst.rxs; }

}
