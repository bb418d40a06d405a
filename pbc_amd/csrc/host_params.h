// host_params.h -- pure host logic of libpbc_hip.so: the pairing object, parameter-text
// parsing and the integer-only part of a_/d_/f_init_pairing.  No HIP calls: shared by the
// library (pbc_hip.hip) and by the host-compiled kernel mirror used in the CPU test-suite
// (tests/hostsim/).
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "fp.cuh"
#include "hostbn.h"
#include "pairing_a.cuh"
#include "pairing_d.cuh"
#include "pairing_f.cuh"
#include "pairing_e.cuh"
#include "group_ops.cuh"

using namespace pbc;

// ---------------------------------------------------------------------------------------
// error plumbing (pbc_error-style: message to stderr is left to the caller)
// ---------------------------------------------------------------------------------------
inline thread_local char g_err[512];     // one per thread for the whole library (several translation units)
inline int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}

// ---------------------------------------------------------------------------------------
// host object
#ifndef PBC_A_WAVE_MAX
#define PBC_A_WAVE_MAX 5120
#endif
#ifndef PBC_A_WAVE2_MAX
#define PBC_A_WAVE2_MAX 1280           // two wavefronts per unit above hip_wave4_max, up to this size (measured cut-overs: profiles/r05_wave_latency.txt)
#endif
#ifndef PBC_A_WAVE4_MAX
#define PBC_A_WAVE4_MAX 768
#endif
// ---------------------------------------------------------------------------------------
#ifndef PBC_D_WAVE_MAX
#define PBC_D_WAVE_MAX 5120
#define PBC_F_WAVE_MAX 4096
#endif
constexpr size_t kProdChunkDefault = (size_t) 1 << 22;   // type a products: terms per launch of the one-term-per-lane kernels unless "hip_prod_chunk N" says otherwise
// the schedules of the wave-per-pairing type d kernels (dw_sched.h): four of them one after the other
struct DwSched {
  std::vector<uint64_t> e;
  size_t off[4] = {0, 0, 0, 0};     // pairing, Miller value of a term, product + final exponentiation, pairing_pp_apply
  int lines = 0;                    // lines of a pairing_pp table
};
struct pbc_hip_pairing_s {
  int type;
  int device;
  int ndev;                  // host-buffer calls: devices the batch is range-split over (0 = just `device`)
  int devs[16];
  int nlimb;                 // 32-bit limbs of F_q
  int deg;                   // types d / g: degree d = k/2 of F_q^d (3 / 5)
  bool a_generic;            // type a outside the 64-byte fast path: runs on the type a1 kernels
  bool a_prod_shared;        // type a fast path, products: "hip_prod_shared 1" keeps one product per lane (a_prod_pairing_lane)
  bool zero_copy;            // host-buffer entry points: kernels read / write pinned caller buffers in place ("hip_zero_copy 0/1")
  bool dynamic;              // resident launches fetch their units from a per-launch counter instead of a fixed stride ("hip_dynamic 1")
  bool no_fair;              // no time-sliced wave priorities (fp.cuh pbc_fair_tick; "hip_no_fair 1")
  bool group_slow;           // group operations: only the complete ladders ("hip_group_slow 1": tests, A/B)
  bool a_multi_compose;      // type a, pow2 / pow3 on G1 / G2: single-base ladders + additions instead of the joint ladder ("hip_multi_compose 1": A/B)
  int resident_slots;        // > 0: workgroups of a resident launch instead of the occupancy query ("hip_resident_slots N", tests)
  size_t host_chunk;         // host-buffer entry points: units per chunk when the parameter text says "hip_host_chunk N" (0: default)
  size_t a_wave4_max;        // ... and up to this size four wavefronts per pairing ("hip_wave4_max N")
  size_t a_wave2_max;        // ... and between the two, up to this size, two ("hip_wave2_max N", round 5)
  std::vector<uint32_t> ag_aux;  // type a1 / type a outside the fast path: the table of the wave kernels (pairing_aw.cuh AG<N>: LEFF, five subtraction constants; empty: this q keeps the lane kernels)
  size_t ag_wave_max;        // ... batches (terms of products) up to this size take them ("hip_wave_max N", 0 = never), four wavefronts per unit up to ag_wave4_max ("hip_wave4_max N")
  size_t ag_wave4_max;
  size_t ag_wave8_max;       // ... and EIGHT (the three-round Miller loop of pairing_aw.cuh miller_loop_p) up to this many terms ("hip_wave8_max N"; type e: never)
  size_t a_wave_max;         // type a fast path, element_pairing: batches up to this size take one WAVEFRONT per pairing (pairing_aw.cuh; "hip_wave_max N", 0 = never)
  size_t d_wave_max;         // type d, five-word fields: single pairings in batches up to this size take one wavefront each (pairing_dw.cuh; "hip_dwave_max N")
  size_t a_prod_chunk;       // ... otherwise: terms per launch of the one-term-per-lane kernels ("hip_prod_chunk N", tests)
  int len_fq, len1, len2, lenT;
#define PBC_HOST_FPK(n) FpK<n> k##n;
  PBC_FOR_EACH_N(PBC_HOST_FPK)  // k5, k6, k7, k16: the one matching nlimb is filled
#undef PBC_HOST_FPK
  AConst a;
  DRaw draw;                 // type D: canonical parameter words for the device-side derivation
  DConst dconst;             // type D: derived tower constants (filled on first use)
  FRaw fraw;                 // type F: canonical parameter words
  FConst fconst;             // type F: derived tower constants (filled on first use)
  FConst fconst_i;           // type F, q = 3 mod 4: the same in the i-basis, for the pairing kernels (pairing_f.cuh init_stage3)
  bool f_bm1;                // fconst_i is valid
  ERaw eraw;                 // type E: integers for the one-time search of the auxiliary point
  EConst econst;             // type E: curve, auxiliary point, exponents (filled on first use)
  bool dev_ready;            // derived constants computed on the device
  bool kargs_checked;        // the constant block's addressing passed its self-test on the device (pbc_hip.hip kargs_selftest)
  int len_zr;                // bytes of a Z_r scalar (pairing_length_in_bytes_Zr)
  uint32_t zr_words[34];     // the group order r (a1: n), little-endian words: the modulus of the Z_r arithmetic (pbc_hip_group2.hip)
  int zr_nlimb;              // smallest built-in field width that holds r (0: none -- r even or wider than 1056 bits)
  double fq_muls_single;     // reference F_q multiplication count per pairing (work model)
  double fq_muls_prod_a, fq_muls_prod_b;   // products: a*k + b
  double fq_muls_pp;         // one pairing_pp_apply (0: no preprocessed variant in the reference)
  // element_from_hash on G1: cofactor and square-root constants (copied into CurveK by fill_curve)
  struct {
    uint32_t cofac[34]; int cofbits;
    int sqrt_mode; uint32_t sqrt_e[34]; int sqrt_bits;
    int ts_s; uint32_t ts_t[34]; int ts_tbits; uint32_t half[34]; int halfbits;
    uint32_t ts_c[34]; bool ts_ready;
  } hash;
  ExtSqrtK xs;               // square roots in the field of the G2 twist (types d, g, f); xs.c derived on first use
  bool xs_ready;
  uint32_t raw_c1[34], raw_c2[34];   // limb-image entry points (pbc_hip.hip raw_prepare): 2^(2 rbits - 64 t) and 2^(64 t) mod q
  int raw_t;                 // ... t = 64-bit limbs of the reference's montfp element (0: constants not derived yet)
  void *counters;            // library only: the unit counters of dynamic resident launches (pbc_hip.hip unit_counter)
  void *host_ctx;            // library only: per-device streams and chunk buffers of the host-buffer path (pbc_hip.hip)
  DwSched gw_sched;                 // type g, five-word field: the schedules of the wave kernels (pbc_hip_d.hip gw_schedules, gw_sched.h)
  DwSched fw_sched;                 // type f, five-word BN fields: the schedules of the wave kernels (pbc_hip_f.hip fw_schedules, fw_sched.h)
  size_t f_wave_max;                // ... single pairings in batches up to this size take it ("hip_fwave_max N")
  DwSched dw_sched;                 // type d, five-word fields: the schedules of the wave kernels (pbc_hip_d.hip dw_schedules, dw_sched.h)
  std::string param_text;    // the parameter text the object was built from (text formats: pbc_hip_param_snprint, host_text.h)
};

// |K*| = q^m - 1 = 2^s T for the twist's field K = F_q^m (m = d for types d / g, 2 for type f)
static int fill_ext_sqrt(pbc_hip_pairing_s *P, const pbc_host::Big &q, int m) {
  using pbc_host::Big;
  memset(&P->xs, 0, sizeof P->xs);
  P->xs_ready = false;
  Big n = q, two, rem;
  two.w.push_back(2);
  for (int i = 1; i < m; i++) n = Big::mul(n, q);
  n.sub_small(1);
  int s = 0;
  while (!n.bit(0)) { n = Big::div(n, two, &rem); s++; }
  if (n.bits() > 24 * 32) return fail("twist field wider than 768 bits");
  n.to_words(P->xs.t, 24);
  P->xs.tbits = n.bits();
  P->xs.s = s;
  Big e = n;
  e.sub_small(1);
  e = Big::div(e, two, &rem);
  e.to_words(P->xs.e, 24);
  P->xs.ebits = e.bits();
  return 0;
}

// curve_from_hash constants: cofactor (0 = none) and the square-root recipe of F_q
static int fill_hash_consts(pbc_hip_pairing_s *P, const pbc_host::Big &q, const pbc_host::Big *cofac) {
  using pbc_host::Big;
  memset(&P->hash, 0, sizeof P->hash);
  if (cofac) {
    if (cofac->bits() > 34 * 32) return fail("cofactor wider than 1088 bits");
    cofac->to_words(P->hash.cofac, 34);
    P->hash.cofbits = cofac->bits();
  }
  Big two, four, rem;
  two.w.push_back(2);
  four.w.push_back(4);
  if ((q.w[0] & 3) == 3) {
    Big e = q;
    e.add_small(1);
    e = Big::div(e, four, &rem);
    e.to_words(P->hash.sqrt_e, 34);
    P->hash.sqrt_bits = e.bits();
    P->hash.sqrt_mode = 0;
    P->hash.ts_ready = true;
    return 0;
  }
  Big qm1 = q;
  qm1.sub_small(1);
  Big half = Big::div(qm1, two, &rem);
  half.to_words(P->hash.half, 34);
  P->hash.halfbits = half.bits();
  Big t = qm1;
  int s = 0;
  while (!t.bit(0)) { t = Big::div(t, two, &rem); s++; }
  t.to_words(P->hash.ts_t, 34);
  P->hash.ts_tbits = t.bits();
  P->hash.ts_s = s;
  Big e = t;
  e.sub_small(1);
  e = Big::div(e, two, &rem);
  e.to_words(P->hash.sqrt_e, 34);
  P->hash.sqrt_bits = e.bits();
  P->hash.sqrt_mode = 1;
  return 0;
}

// min_bits: smallest modulus accepted for this word count (default: the top word is in use)
template <int N>
static int fill_fpk(FpK<N> &K, const pbc_host::Big &q, int min_bits = 32 * (N - 1) + 1) {
  using pbc_host::Big;
  if (q.bits() > 32 * N || q.bits() < min_bits || !(q.w[0] & 1)) return 1;
  memset(&K, 0, sizeof K);
  q.to_words(K.p, N);
  const int rbits = Limbs29<N>::W * Limbs29<N>::L;
  Big::pow2_mod(rbits, q).to_words(K.one, N);
  Big::pow2_mod(2 * rbits, q).to_words(K.r2, N);
  for (int i = 0; i < Limbs29<N>::L; i++) {
    uint32_t v = 0;
    for (int b = 0; b < Limbs29<N>::W; b++) v |= (uint32_t) q.bit(Limbs29<N>::W * i + b) << b;
    K.p29[i] = v;
  }
  K.ninv29 = pbc_host::neg_inv32(K.p[0]) & Limbs29<N>::MASK;
  Big::pow2_mod(3 * rbits, q).to_words(K.r3, N);
  for (int i = 0; i < Inv30<N>::L; i++) {
    uint32_t v = 0;
    for (int b = 0; b < 30; b++) v |= (uint32_t) q.bit(30 * i + b) << b;
    K.p30[i] = v;
  }
  K.qinv30 = (0u - pbc_host::neg_inv32(K.p[0])) & 0x3fffffffu;
  Big pm2 = q;
  pm2.sub_small(2);
  pm2.to_words(K.pm2, N);
  K.ninv = pbc_host::neg_inv32(K.p[0]);
  K.pbits = (uint32_t) q.bits();
  K.fbytes = (uint32_t) ((q.bits() + 7) / 8);
  return 0;
}

// Z_r: byte length of a scalar and the modulus of the batched Z_r arithmetic (the reference runs its F_p back end on r:
// pairing->Zr, ecc/pairing.c / field_init_fp)
static void set_zr(pbc_hip_pairing_s *P, const pbc_host::Big &r) {
  P->len_zr = (r.bits() + 7) / 8;
  memset(P->zr_words, 0, sizeof P->zr_words);
  P->zr_nlimb = 0;
  if (r.bits() > 33 * 32 || r.bits() < 2 || !(r.w[0] & 1)) return;
  r.to_words(P->zr_words, 34);
  static const int widths[6] = {5, 6, 7, 8, 16, 33};
  for (int i = 0; i < 6 && !P->zr_nlimb; i++)
    if (r.bits() <= 32 * widths[i]) P->zr_nlimb = widths[i];
}

// The wave kernels of type a1 and of type a parameters outside the fast path (pairing_aw.cuh, AG<N>): q must leave ten bits of
// the radix 2^(W L) free (the value bounds of the limb-form routines) and fill at least twelve bits of its top limb (the
// borrowed subtraction constants: what the bound tracker of pairing_al.cuh assumes for a.param's q >= 2^504); the table holds
// LEFF = the limbs q fills and c q in borrowed form for (c, D) = (2, 1) (4, 2) (8, 4) (12, 2) (16, 2).  "hip_wave_max 0": never.
template <int N>
static void ag_aux_build(pbc_hip_pairing_s *P, const pbc_host::Big &q, const char *txt, size_t len, bool type_e = false) {
  constexpr int W = Limbs29<N>::W, L = Limbs29<N>::L;
  P->ag_aux.clear();
  // measured cut-overs (profiles/r06_agwave_latency.txt): a lane needs 0.2 s (a1.param) / 38 ms (a_160_1024) / 6.5 ms (512-bit q) however
  // small the batch; the wave kernels saturate at 72 k / 330 k / 1.0 M pairings a second.  On the 33-word fields four wavefronts per
  // unit are the faster shape at every size (two 256-register waves fit a SIMD either way)
  int wave_max = N >= 32 ? 12288 : 6144, wave4_max = N >= 32 ? 12288 : 1024;
  // type e (pairing_ew.cuh, profiles/r06_ewave_latency.txt): a third of a pairing is the (q - 1) / r power, one product at a time --
  // one wavefront per unit carries twice the units a second (300 k against 135 k on e.param; a lane: 34.5 ms for any batch)
  if (type_e && N >= 32) { wave_max = 10240; wave4_max = 512; }
  pbc_host::param_int(txt, len, "hip_wave_max", wave_max);
  pbc_host::param_int(txt, len, "hip_wave4_max", wave4_max);
  P->ag_wave_max = wave_max < 0 ? 0 : (size_t) wave_max;
  P->ag_wave4_max = wave4_max < 0 ? 0 : (size_t) wave4_max;
  // OFF unless the parameter text asks ("hip_wave8_max N"): a1.param 7.7 ms a pairing against 9.1 with four wavefronts (up to 256 units), same
  // bytes -- but with the route on by default the GPU suite aborted in a LATER test of the same process in 2 of 7 runs (12 of 12 clean with it
  // off; profiles/r06_notes.md): not understood, so not shipped as the default
  int wave8_max = 0;
  pbc_host::param_int(txt, len, "hip_wave8_max", wave8_max);
  P->ag_wave8_max = wave8_max < 0 || type_e ? 0 : (size_t) wave8_max;
  const int qb = q.bits(), leff = (qb + W - 1) / W, top = qb - W * (leff - 1);
  if (leff > L || leff < 3 || W * L - qb < 10 || top < 12) return;
  std::vector<uint32_t> a(pbc::AW_AUX_HEAD + 5 * L, 0u);
  a[0] = (uint32_t) leff;
  static const uint32_t cd[5][2] = {{2, 1}, {4, 2}, {8, 4}, {12, 2}, {16, 2}};
  for (int t = 0; t < 5; t++)
    if (pbc_host::ksub_build(q, leff, W * (leff - 1) + 12, cd[t][0], cd[t][1], &a[pbc::AW_AUX_HEAD + t * L], W)) return;
  P->ag_aux.swap(a);
}
// a_init_pairing (ecc/a_param.c:1431-1472) + pbc_param_init_a (:1489-1502)
static int init_type_a(pbc_hip_pairing_s *P, const char *txt, size_t len) {
  using namespace pbc_host;
  Big q, r, h;
  int exp2, exp1, sign1, sign0;
  if (!param_big(txt, len, "q", q) || !param_big(txt, len, "r", r) || !param_big(txt, len, "h", h) ||
      !param_int(txt, len, "exp2", exp2) || !param_int(txt, len, "exp1", exp1) ||
      !param_int(txt, len, "sign1", sign1) || !param_int(txt, len, "sign0", sign0))
    return fail("type a: missing q/r/h/exp2/exp1/sign1/sign0");
  if (exp1 <= 0 || exp2 <= exp1) return fail("type a: bad exp1/exp2");
  if ((q.w[0] & 3) != 3) return fail("type a: q must be 3 mod 4");
  if (h.is_zero() || h.bits() > 34 * 32) return fail("type a: bad cofactor");
  {
    Big qp1 = q;
    qp1.add_small(1);
    if (Big::cmp(Big::mul(r, h), qp1) != 0) return fail("type a: q + 1 != r h");
  }
  memset(&P->a, 0, sizeof P->a);
  h.to_words(P->a.h, 34);
  P->a.hbits = h.bits();
  {
    Big e = q, four, rem;
    e.add_small(1);
    four.w.push_back(4);
    e = Big::div(e, four, &rem);
    e.to_words(P->a.sqrt_e, 34);
    P->a.sqrt_bits = e.bits();
  }
  P->a.exp2 = exp2;
  P->a.exp1 = exp1;
  P->a.sign1 = sign1;
  P->len_fq = (q.bits() + 7) / 8;
  P->len1 = P->len2 = P->lenT = 2 * P->len_fq;
  set_zr(P, r);
  // the standard size (pbc_param_init_a_gen(160, 512), a.param): dedicated kernels -- Solinas loop with its single
  // addition, 16-byte vector loads/stores of the 128-byte records, F_q in limb form.  The limb-form kernel's
  // subtraction constants (AConst::ksub) borrow from q's top 29-bit limb: checked here for this q (hostbn.h
  // ksub_build; the bound tracker of pairing_al.cuh assumes q >= 2^504); a q that fails runs on the generic kernels.
  bool fast = P->len_fq == 64;
  if (fast) {
    static const uint32_t cd[5][2] = {{2, 1}, {4, 2}, {8, 4}, {12, 2}, {16, 2}};
    for (int t = 0; t < 5; t++)
      if (pbc_host::ksub_build(q, 18, 505, cd[t][0], cd[t][1], P->a.ksub[t])) fast = false;
  }
  if (fast) {
    if (fill_fpk<16>(P->k16, q)) return fail("type a: bad q");
    P->nlimb = 16;
    P->a_generic = false;
    int shared = 0;                    // products: one term per lane (pairing_al.cuh) unless the parameter text says otherwise
    param_int(txt, len, "hip_prod_shared", shared);
    P->a_prod_shared = shared != 0;
    int chunk = (int) kProdChunkDefault;   // 2^22 terms = 640 MB of 160-byte workspace records
    param_int(txt, len, "hip_prod_chunk", chunk);
    P->a_prod_chunk = chunk < 1 ? 1 : (size_t) chunk;
    int wave_max = PBC_A_WAVE_MAX;     // the batch size below which a wavefront per pairing is the faster launch (bench.py --sweep)
    param_int(txt, len, "hip_wave_max", wave_max);
    P->a_wave_max = wave_max < 0 ? 0 : (size_t) wave_max;
    int wave4_max = PBC_A_WAVE4_MAX;
    param_int(txt, len, "hip_wave4_max", wave4_max);
    P->a_wave4_max = wave4_max < 0 ? 0 : (size_t) wave4_max;
    int wave2_max = PBC_A_WAVE2_MAX;
    param_int(txt, len, "hip_wave2_max", wave2_max);
    P->a_wave2_max = wave2_max < 0 ? 0 : (size_t) wave2_max;
  } else {
    // any other size up to 1056 bits: the type a1 kernels (plain double-and-add over the bits of r;
    // functions with the same divisor up to vertical lines, which the final power removes) on the
    // 16- or 33-word arithmetic
    if (q.bits() < 160 || r.bits() > 34 * 32 - 1 || r.bits() < 3)
      return fail("type a: only 160..1056-bit q is supported by this build (got %d bits)", q.bits());
    if (q.bits() <= 512 ? fill_fpk<16>(P->k16, q, 160) : fill_fpk<33>(P->k33, q, 513))
      return fail("type a: only 160..1056-bit q is supported by this build (got %d bits)", q.bits());
    P->nlimb = q.bits() <= 512 ? 16 : 33;
    P->a_generic = true;
    if (P->nlimb == 16) ag_aux_build<16>(P, q, txt, len); else ag_aux_build<33>(P, q, txt, len);
    P->a.rbits = pbc_host::naf_of_half(r, P->a.r, P->a.rm, 34);      // signed digits: three for a Solinas r
    if (!P->a.rbits) return fail("type a: r too wide for the Miller loop digits");
  }
  // work model: SURVEY.md 8d instrumented the reference on a.param (exp2 = 159, 353-bit h): 3675 F_q
  // products in a_pairing_proj's Miller loop + 717 in a_tateexp; a_pairings_affine (a_param.c:1283-1383)
  // 41377 for k = 16 (linear model 2543 k + 689); a_pairing_pp_apply 1838.
  // Other sizes scale with the loop lengths exp2 and bits(h).
  const double sm = exp2 / 159.0, sh = h.bits() / 353.0;
  P->fq_muls_single = 3675.0 * sm + 717.0 * sh;
  P->fq_muls_prod_a = 2543.0 * sm;
  P->fq_muls_prod_b = 689.0 * sh;
  P->fq_muls_pp = 1121.0 * sm + 717.0 * sh;
  if (fill_hash_consts(P, q, &h)) return 1;   // field_init_curve_ab(Eq, a, b, r, h): cofactor h (a_param.c:1453)
  return 0;
}

// a1_init_pairing (ecc/a_param.c:2230-2273) + pbc_param_init_a1 (:2289-2298): y^2 = x^3 + x over F_p,
// group order n (composite), cofactor l = (p + 1)/n
static int init_type_a1(pbc_hip_pairing_s *P, const char *txt, size_t len) {
  using namespace pbc_host;
  Big p, n, l;
  if (!param_big(txt, len, "p", p) || !param_big(txt, len, "n", n) || !param_big(txt, len, "l", l))
    return fail("type a1: missing p/n/l");
  // a1.param is 1033 bits (33 words); smaller orders run on the 16-word arithmetic
  if (p.bits() <= 512 ? fill_fpk<16>(P->k16, p, 160) : fill_fpk<33>(P->k33, p, 513))
    return fail("type a1: only 160..1056-bit p is supported by this build (got %d bits)", p.bits());
  if ((p.w[0] & 3) != 3) return fail("type a1: p must be 3 mod 4");
  {
    Big pp1 = p;
    pp1.add_small(1);
    if (Big::cmp(Big::mul(n, l), pp1) != 0) return fail("type a1: p + 1 != n l");
  }
  if (l.bits() > 512 || l.is_zero() || n.bits() > 34 * 32 || n.bits() < 3) return fail("type a1: bad n or l");
  memset(&P->a, 0, sizeof P->a);
  l.to_words(P->a.h, 16);
  P->a.hbits = l.bits();
  P->a.rbits = pbc_host::naf_of_half(n, P->a.r, P->a.rm, 34);
  if (!P->a.rbits) return fail("type a1: n too wide for the Miller loop digits");
  {
    Big e = p, four, rem;
    e.add_small(1);
    four.w.push_back(4);
    e = Big::div(e, four, &rem);
    e.to_words(P->a.sqrt_e, 34);
    P->a.sqrt_bits = e.bits();
  }
  P->nlimb = p.bits() <= 512 ? 16 : 33;
  P->len_fq = (p.bits() + 7) / 8;
  P->len1 = P->len2 = P->lenT = 2 * P->len_fq;
  set_zr(P, n);
  if (P->nlimb == 16) ag_aux_build<16>(P, p, txt, len); else ag_aux_build<33>(P, p, txt, len);
  // work model (a1_pairing_proj, a_param.c:1840-2015): per bit of n a tangent (10 F_p products incl.
  // the projective line), a doubling (10) and an F_p^2 square + product (2 + 3); per set bit a chord
  // (8), a mixed addition (12) and a product (3); f^(p-1) and the 11-bit power are negligible.
  int ones = 0;
  for (int i = 1; i < n.bits() - 1; i++) ones += n.bit(i);
  P->fq_muls_single = 25.0 * (n.bits() - 1) + 23.0 * ones;
  P->fq_muls_prod_a = P->fq_muls_single;
  P->fq_muls_prod_b = 0.0;
  // a1_pairing_pp_apply (a_param.c:1728-1818): per step an F_p^2 square and product (5) plus the
  // evaluation of the stored line (2), per set bit another evaluation and product (5)
  P->fq_muls_pp = 7.0 * (n.bits() - 1) + 5.0 * ones;
  return fill_hash_consts(P, p, &l);     // cofactor phikonr = l (a_param.c:2250)
}

// e_init_pairing (ecc/e_param.c:832-872) + pbc_param_init_e (:891-906): host part (integers only).
// The 1020-bit field of e.param runs on the 33-word arithmetic built for type a1.
static int init_type_e(pbc_hip_pairing_s *P, const char *txt, size_t len) {
  using namespace pbc_host;
  Big q, r, a, b;
  if (!param_big(txt, len, "q", q) || !param_big(txt, len, "r", r) || !param_big(txt, len, "a", a) ||
      !param_big(txt, len, "b", b))
    return fail("type e: missing q/r/a/b");
  if (q.bits() <= 512 ? fill_fpk<16>(P->k16, q, 160) : fill_fpk<33>(P->k33, q, 513))
    return fail("type e: only odd 160..1056-bit q is supported by this build (got %d bits)", q.bits());
  if (Big::cmp(a, q) >= 0 || Big::cmp(b, q) >= 0) return fail("type e: coefficient >= q");
  if (r.bits() > 256 || r.bits() < 3 || !(r.w[0] & 1)) return fail("type e: bad r");
  memset(&P->eraw, 0, sizeof P->eraw);
  memset(&P->econst, 0, sizeof P->econst);
  a.to_words(P->eraw.a, NE_MAX);
  b.to_words(P->eraw.b, NE_MAX);
  Big two, rem;
  two.w.push_back(2);
  Big qm1 = q;
  qm1.sub_small(1);
  Big half = Big::div(qm1, two, &rem);
  half.to_words(P->eraw.half, NE_MAX);
  P->eraw.halfbits = half.bits();
  Big t = qm1;
  int s = 0;
  while (!t.bit(0)) { t = Big::div(t, two, &rem); s++; }
  t.to_words(P->eraw.t, NE_MAX);
  P->eraw.tbits = t.bits();
  P->eraw.s = s;
  Big t1 = t;
  t1.add_small(1);
  t1 = Big::div(t1, two, &rem);
  t1.to_words(P->eraw.t1h, NE_MAX);
  P->eraw.t1hbits = t1.bits();
  // phikonr = (q - 1)/r (k = 1; e_param.c:857-860)
  Big phik = Big::div(qm1, r, &rem);
  if (!rem.is_zero()) return fail("type e: r does not divide q - 1");
  phik.to_words(P->econst.phik, NE_MAX);
  P->econst.phikbits = phik.bits();
  r.to_words(P->econst.r, 8);
  P->econst.rbits = r.bits();
  P->nlimb = q.bits() <= 512 ? 16 : 33;
  P->len_fq = (q.bits() + 7) / 8;
  P->len1 = P->len2 = 2 * P->len_fq;
  P->lenT = P->len_fq;
  if (P->nlimb == 16) ag_aux_build<16>(P, q, txt, len, true); else ag_aux_build<33>(P, q, txt, len, true);   // small batches: pairing_ew.cuh
  set_zr(P, r);
  // work model (e_miller_proj, e_param.c:64-300, + element_pow_mpz by (q-1)/r): per doubling about
  // 2 squarings + tangent (8, two evaluation points 3 each) + Jacobian doubling (10) + verticals (2)
  // + 4 accumulations = 32 F_q products; the final power is 1.5 products per exponent bit.
  P->fq_muls_single = 32.0 * (r.bits() - 1) + 1.5 * phik.bits();
  P->fq_muls_prod_a = P->fq_muls_single;   // generic_prod_pairings: k full pairings
  P->fq_muls_prod_b = 0.0;
  Big h;
  if (!param_big(txt, len, "h", h)) return fail("type e: missing h");
  return fill_hash_consts(P, q, &h);       // cofactor h (e_param.c:853)
}

// d_init_pairing (ecc/d_param.c:993-1095) + pbc_param_init_d, and g_init_pairing (ecc/g_param.c:
// 1248-1354) + pbc_param_init_g (:1378-1402): host part (integers only); the tower constants are
// derived on the device at first use (pairing_d.cuh init_stage*).  deg = k/2: 3 (type d), 5 (type g).
static int init_type_d(pbc_hip_pairing_s *P, const char *txt, size_t len, int deg) {
  using namespace pbc_host;
  const char *tn = deg == 3 ? "type d" : "type g";
  Big q, r, a, b, nqr, co[DEG_MAX];
  int k = 0;
  if (!param_big(txt, len, "q", q) || !param_big(txt, len, "r", r) || !param_big(txt, len, "a", a) ||
      !param_big(txt, len, "b", b) || !param_big(txt, len, "nqr", nqr) || !param_int(txt, len, "k", k))
    return fail("%s: missing q/r/a/b/k/nqr", tn);
  for (int i = 0; i < deg; i++) {
    char key[8];
    snprintf(key, sizeof key, "coeff%d", i);
    if (!param_big(txt, len, key, co[i])) return fail("%s: missing %s", tn, key);
  }
  if (k != 2 * deg) return fail("%s: only embedding degree %d is supported (got %d)", tn, 2 * deg, k);
  // field width: 5 words for d159 and g149, 6 for d277699-175-167 / d278027-190-181, 7 for
  // d105171-196-185 / d201 / d224 (all the type d and type g files under param/)
  const int ND = (q.bits() + 31) / 32;
  if (ND < 5 || ND > ND_MAX || (ND == 5 ? fill_fpk<5>(P->k5, q) : ND == 6 ? fill_fpk<6>(P->k6, q) : fill_fpk<7>(P->k7, q)))
    return fail("%s: only odd 129..224-bit q is supported by this build (got %d bits)", tn, q.bits());
  if (deg == 5 && ND != 5) return fail("type g: only 129..160-bit q is supported by this build (got %d bits)", q.bits());
  if (Big::cmp(a, q) >= 0 || Big::cmp(b, q) >= 0 || Big::cmp(nqr, q) >= 0) return fail("%s: coefficient >= q", tn);
  memset(&P->draw, 0, sizeof P->draw);
  memset(&P->dconst, 0, sizeof P->dconst);
  a.to_words(P->draw.a, ND);
  b.to_words(P->draw.b, ND);
  nqr.to_words(P->draw.nqr, ND);
  for (int i = 0; i < deg; i++) {
    if (Big::cmp(co[i], q) >= 0) return fail("%s: coefficient >= q", tn);
    co[i].to_words(P->draw.coeff[i], ND);
  }
  q.to_words(P->draw.q, ND + 1);
  P->draw.qbits = q.bits();
  if (r.bits() > 256 || r.bits() < 3) return fail("%s: bad r", tn);
  {
    // small batches of single pairings on the five-word d = 3 fields: one pairing per WAVEFRONT (pairing_dw.cuh) up to this
    // batch size ("hip_dwave_max N", 0 = never); above it the one-pairing-per-lane kernel is the faster launch
    int dwave_max = PBC_D_WAVE_MAX;
    param_int(txt, len, "hip_dwave_max", dwave_max);
    P->d_wave_max = dwave_max < 0 ? 0 : (size_t) dwave_max;
  }
  {
    // limb-form point arithmetic of the 5-word d = 3 kernels (pairing_d.cuh, kLimbPoint): its subtraction constants borrow
    // from q's top 29-bit limb, which must hold at least 8 bits of q; "hip_no_limb 1" forces the word-form routines
    int no_limb = 0;
    param_int(txt, len, "hip_no_limb", no_limb);
    P->dconst.limb_ok = (ND == 5 && deg == 3 && !no_limb) ? 1 : 0;
    if (P->dconst.limb_ok) {             // the constants d_init_lane builds on the device, checked for this q (hostbn.h)
      static const uint32_t cd[4][2] = {{2, 1}, {4, 2}, {16, 2}, {32, 2}};
      uint32_t k[6];
      for (int t = 0; t < 4; t++)
        if (pbc_host::ksub_build(q, 6, 29 * 5 + 8, cd[t][0], cd[t][1], k)) P->dconst.limb_ok = 0;
    }
  }
  P->dconst.rbits = pbc_host::naf_of_half(r, P->dconst.r, P->dconst.rm, 9);     // signed digits of the Miller loop
  if (!P->dconst.rbits) return fail("%s: r too wide for the Miller loop digits", tn);
  // phikonr = Phi_k(q)/r: (q^2 - q + 1)/r (d_param.c:1036-1042), (q^4 - q^3 + q^2 - q + 1)/r (g_param.c:1288-1305)
  Big q2 = Big::mul(q, q);
  Big z = q2;
  z.sub(q);
  z.add_small(1);
  if (deg == 5) {
    Big q3 = Big::mul(q2, q), q4 = Big::mul(q2, q2);
    z = Big::add(z, q4);
    z.sub(q3);
  }
  Big rem;
  Big phik = Big::div(z, r, &rem);
  if (!rem.is_zero() || phik.bits() > 512) return fail("%s: r does not divide Phi_k(q)", tn);
  phik.to_words(P->dconst.phik, 16);
  P->dconst.phikbits = phik.bits();
  P->nlimb = ND;
  P->deg = deg;
  P->len_fq = (q.bits() + 7) / 8;
  P->len1 = 2 * P->len_fq;
  P->len2 = P->lenT = 2 * deg * P->len_fq;
  set_zr(P, r);
  if (deg == 3) {
    // work model: SURVEY.md 8d instrumented the reference on d159.param (158-bit r, 161-bit
    // (q^2-q+1)/r): 22254 F_q products in the Miller loop + 4197 in cc_tatepower.  Both loops are
    // one iteration per exponent bit, so other parameter files scale by their bit lengths.
    const double miller = 22254.0 * r.bits() / 158.0, tate = 4197.0 * phik.bits() / 161.0;
    P->fq_muls_single = miller + tate;
    P->fq_muls_prod_a = miller;            // per-term Miller work + one cc_tatepower
    P->fq_muls_prod_b = tate;
    // d_pairing_pp_apply (d_param.c:908-966) skips the affine point arithmetic of the Miller loop:
    // tangent coefficients + doubling (7 products per bit of r), chord + addition (5 per set bit)
    int adds = 0;
    for (int i = 1; i < r.bits() - 1; i++) adds += r.bit(i);
    P->fq_muls_pp = P->fq_muls_single - 7.0 * (r.bits() - 1) - 5.0 * adds;
  } else {
    // g149.param (149-bit r, 447-bit Phi_10(q)/r): F_q products counted on the CPU restatement of
    // the reference's algorithm (the test oracle's counters; polymod_mul is schoolbook + table for
    // d = 5, 45 F_q products): 96795 per pairing, about 55100 in the Miller loop and 41695 in tatepower10.
    const double miller = 55100.0 * r.bits() / 149.0, tate = 41695.0 * phik.bits() / 447.0;
    P->fq_muls_single = miller + tate;
    P->fq_muls_prod_a = miller + tate;     // generic_prod_pairings (pairing.c:35-46): k full pairings
    P->fq_muls_prod_b = 0.0;
    int adds = 0;                          // g_pairing_pp_apply (g_param.c:741-787), as for type d
    for (int i = 1; i < r.bits() - 1; i++) adds += r.bit(i);
    P->fq_muls_pp = P->fq_muls_single - 7.0 * (r.bits() - 1) - 5.0 * adds;
  }
  Big h;
  if (!param_big(txt, len, "h", h)) return fail("%s: missing h", tn);
  if (fill_ext_sqrt(P, q, deg)) return 1;  // G2 = E'(F_q^d)
  return fill_hash_consts(P, q, &h);       // cofactor h (d_param.c:1016, g_param.c:1267)
}

// f_init_pairing (ecc/f_param.c:335-447): host part (integers only)
static int init_type_f(pbc_hip_pairing_s *P, const char *txt, size_t len) {
  using namespace pbc_host;
  Big q, r, b, beta, a0, a1;
  if (!param_big(txt, len, "q", q) || !param_big(txt, len, "r", r) || !param_big(txt, len, "b", b) ||
      !param_big(txt, len, "beta", beta) || !param_big(txt, len, "alpha0", a0) || !param_big(txt, len, "alpha1", a1))
    return fail("type f: missing q/r/b/beta/alpha0/alpha1");
  // f.param has a 158-bit q (5 words); pbc_param_init_f_gen(bits) up to 256 bits runs on 8 words
  const int NF = q.bits() <= 160 ? 5 : 8;
  if (NF == 5 ? fill_fpk<5>(P->k5, q) : fill_fpk<8>(P->k8, q, 161))
    return fail("type f: only 129..256-bit q is supported by this build (got %d bits)", q.bits());
  if (Big::cmp(b, q) >= 0 || Big::cmp(beta, q) >= 0 || Big::cmp(a0, q) >= 0 || Big::cmp(a1, q) >= 0)
    return fail("type f: coefficient >= q");
  memset(&P->fraw, 0, sizeof P->fraw);
  memset(&P->fconst, 0, sizeof P->fconst);
  b.to_words(P->fraw.b, NF);
  beta.to_words(P->fraw.beta, NF);
  a0.to_words(P->fraw.alpha0, NF);
  a1.to_words(P->fraw.alpha1, NF);
  // (q - 1)/6: X^q = negalpha^((q-1)/6) X
  Big qm1 = q, six, rem;
  qm1.sub_small(1);
  six.w.push_back(6);
  Big e6 = Big::div(qm1, six, &rem);
  if (!rem.is_zero()) return fail("type f: q must be 1 mod 6");
  e6.to_words(P->fraw.e6, NF + 1);
  P->fraw.e6bits = e6.bits();
  {
    // q = 3 mod 4: the pairing kernels work in the i-basis of F_q^2 (pairing_f.cuh init_stage3); "hip_no_bm1 1" keeps the
    // parameter file's beta (tests).  K = 4 q in borrowed limbs must dominate normalised values below 2.001 q.
    int no_bm1 = 0;
    param_int(txt, len, "hip_no_bm1", no_bm1);
    P->f_bm1 = false;
    const int L = NF == 5 ? Limbs29<5>::L : Limbs29<8>::L;
    // borrowed multiples of q for differences in limb form: 4 q (D = 1) dominates a normalised value below 2.001 q,
    // 8 q (D = 2) its double (the cyclotomic squaring); both need q to reach two bits into its top limb
    P->fraw.k_ok = !pbc_host::ksub_build(q, L, 29 * (L - 1) + 2, 4, 1, P->fraw.kneg29) &&
                   !pbc_host::ksub_build(q, L, 29 * (L - 1) + 2, 8, 2, P->fraw.kneg8_29);
    {
      // the limb-form steps on E(F_q) (pairing_f.cuh f_dbl_core_l): 2 q, 4 q, 16 q, 32 q in borrowed limbs, six-limb fields
      // whose q reaches eight bits into its top limb (the bound tracker's assumptions); "hip_no_limb 1" keeps the word form
      static const uint32_t cd[4][2] = {{2, 1}, {4, 2}, {16, 2}, {32, 2}};
      int no_limb = 0;
      param_int(txt, len, "hip_no_limb", no_limb);
      P->fraw.pl_ok = (L == 6 && !no_limb) ? 1 : 0;
      for (int t = 0; t < 4 && P->fraw.pl_ok; t++)
        if (pbc_host::ksub_build(q, 6, 29 * 5 + 8, cd[t][0], cd[t][1], P->fraw.pk29[t])) P->fraw.pl_ok = 0;
    }
    int no_cyc = 0;
    param_int(txt, len, "hip_no_cyc", no_cyc);   // tests: plain squarings in the hard part
    if ((q.w[0] & 3) == 3 && !no_bm1 && P->fraw.k_ok) {
      Big e4 = q, four;
      e4.add_small(1);
      four.w.push_back(4);
      e4 = Big::div(e4, four, &rem);
      e4.to_words(P->fraw.e4, NF + 1);
      P->fraw.e4bits = e4.bits();
    }
    if (no_cyc) P->fraw.k_ok = 0;
    // A sparse xi for the pairing kernels (pairing_f.cuh init_stage4): the integers of the 6th-root computation in F_q^2.
    // |F_q^2*| = q^2 - 1 = S m with S = 2^a 3^b and gcd(m, 6) = 1.  For w a 6th power: w = w_S w_m with
    // w_S = w^(m u), u = 1/m mod S (in the subgroup of order S, generated by xi^m: xi is neither a square nor a cube) and
    // w_m = w^(S v), v = 1/S mod m, whose 6th root is w_m^t, t = 1/6 mod m.  "hip_no_xs 1" keeps the parameter file's xi.
    int no_xs = 0;
    param_int(txt, len, "hip_no_xs", no_xs);
    P->fraw.xs_try = 0;
    if (P->fraw.e4bits > 0 && L == 6 && P->fraw.k_ok && !no_xs) {
      Big N = Big::mul(q, q), two, three, rm2;
      N.sub_small(1);
      two.w.push_back(2);
      three.w.push_back(3);
      Big m = N;
      uint32_t S = 1;
      for (;;) { Big t = Big::div(m, two, &rm2); if (!rm2.is_zero()) break; m = t; S *= 2; }
      for (;;) { Big t = Big::div(m, three, &rm2); if (!rm2.is_zero()) break; m = t; S *= 3; }
      auto mod_small = [](const Big &a, uint32_t d) { Big dd, rr; dd.w.push_back(d); (void) Big::div(a, dd, &rr); return rr.is_zero() ? 0u : rr.w[0]; };
      auto times = [](const Big &a, uint32_t k) { Big kk; kk.w.push_back(k); return Big::mul(a, kk); };
      if (S < (1u << 20)) {
        const uint32_t mS = mod_small(m, S);
        uint32_t u = 0, kv = 0, kt = 0;
        for (uint32_t x = 1; x < S; x++) if ((uint64_t) x * mS % S == 1) u = x;                 // u = 1/m mod S
        for (uint32_t x = 0; x < S; x++) if (((uint64_t) x * mS + 1) % S == 0) kv = x;          // S | m kv + 1
        const uint32_t m6 = mod_small(m, 6);
        for (uint32_t x = 0; x < 6; x++) if ((x * m6 + 1) % 6 == 0) kt = x;                     // 6 | m kt + 1
        Big sv, six2, dS, v, t;
        six2.w.push_back(6);
        dS.w.push_back(S);
        sv = times(m, kv); sv.add_small(1); v = Big::div(sv, dS, &rm2);                         // v = 1/S mod m
        sv = times(m, kt); sv.add_small(1); t = Big::div(sv, six2, &rm2);                       // t = 1/6 mod m
        Big eS = times(m, u), em = Big::mul(times(v, S), t), ecls = Big::div(N, six2, &rm2);
        (void) Big::div(em, N, &rm2);
        em = rm2;
        if (u && m.bits() <= 352 && eS.bits() <= 352 && em.bits() <= 352 && ecls.bits() <= 352) {
          m.to_words(P->fraw.xs_m, 11); P->fraw.xs_mbits = m.bits();
          eS.to_words(P->fraw.xs_eS, 11); P->fraw.xs_eSbits = eS.bits();
          em.to_words(P->fraw.xs_em, 11); P->fraw.xs_embits = em.bits();
          ecls.to_words(P->fraw.xs_ecls, 11); P->fraw.xs_eclsbits = ecls.bits();
          P->fraw.xs_S = (int) S;
          P->fraw.xs_try = 1;
        }
      }
    }
  }
  if (r.bits() > 256 || r.bits() < 3) return fail("type f: bad r");
  P->fconst.rbits = pbc_host::naf_of_half(r, P->fconst.r, P->fconst.rm, 9);     // signed digits of the Miller loop
  if (!P->fconst.rbits) return fail("type f: r too wide for the Miller loop digits");
  // tateexp = ((q^2 - 1) q^2 + 1)/r (f_param.c:414-420)
  Big q2 = Big::mul(q, q), z = q2;
  z.sub_small(1);
  z = Big::mul(z, q2);
  z.add_small(1);
  Big te = Big::div(z, r, &rem);
  if (!rem.is_zero() || te.bits() > 1024) return fail("type f: r does not divide q^4 - q^2 + 1");
  te.to_words(P->fconst.tateexp, 32);
  P->fconst.tebits = te.bits();
  // Recognise the BN family (genfparam / f_param.c:70-95): q = 36x^4 + 36x^3 + 24x^2 + 6x + 1
  // with x of either sign, r = 36x^4 + 36x^3 + 18x^2 + 6x + 1.  Binary search on |x|.
  {
    auto poly = [](uint64_t ax, int sign, unsigned c2) {      // 36x^4 +- 36x^3 + c2 x^2 +- 6x + 1
      Big x, t, acc;
      x.w = {(uint32_t) ax, (uint32_t) (ax >> 32)};
      x.trim();
      Big x2 = Big::mul(x, x), x3 = Big::mul(x2, x), x4 = Big::mul(x2, x2);
      auto times = [](const Big &a, uint32_t k) { Big kk; kk.w.push_back(k); return Big::mul(a, kk); };
      auto add = [](Big a, const Big &b) {
        uint64_t c = 0;
        if (a.w.size() < b.w.size()) a.w.resize(b.w.size(), 0);
        for (size_t i = 0; i < a.w.size(); i++) { c += (uint64_t) a.w[i] + (i < b.w.size() ? b.w[i] : 0); a.w[i] = (uint32_t) c; c >>= 32; }
        if (c) a.w.push_back((uint32_t) c);
        return a;
      };
      acc = add(times(x4, 36), times(x2, c2));
      acc.add_small(1);
      Big odd = add(times(x3, 36), times(x, 6));
      if (sign > 0) acc = add(acc, odd); else acc.sub(odd);
      return acc;
    };
    P->fconst.bn_ok = 0;
    int no_bn = 0;                       // "hip_no_bn 1" in the parameter text forces the generic power
    param_int(txt, len, "hip_no_bn", no_bn);
    for (int sign = 1; sign >= -1 && !P->fconst.bn_ok && !no_bn; sign -= 2) {
      uint64_t lo = 2, hi = (uint64_t) 1 << 62;
      // q ~ 36 x^4  ->  x < 2^((bits+3)/4)
      hi = (uint64_t) 1 << ((q.bits() + 3) / 4);
      while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (Big::cmp(poly(mid, sign, 24), q) < 0) lo = mid + 1; else hi = mid;
      }
      if (Big::cmp(poly(lo, sign, 24), q) == 0 && Big::cmp(poly(lo, sign, 18), r) == 0) {
        P->fconst.bn_ok = 1;
        P->fconst.bn_xneg = sign < 0;
        P->fconst.bn_x[0] = (uint32_t) lo;
        P->fconst.bn_x[1] = (uint32_t) (lo >> 32);
        int nb = 0;
        for (uint64_t t = lo; t; t >>= 1) nb++;
        P->fconst.bn_xbits = nb;
      }
    }
  }
  {
    // small batches of single pairings on the five-word BN fields: one pairing per WAVEFRONT (pairing_fw.cuh) up to this batch
    // size ("hip_fwave_max N", 0 = never)
    int fwave_max = PBC_F_WAVE_MAX;
    param_int(txt, len, "hip_fwave_max", fwave_max);
    P->f_wave_max = fwave_max < 0 ? 0 : (size_t) fwave_max;
  }
  P->nlimb = NF;
  P->len_fq = (q.bits() + 7) / 8;
  P->len1 = 2 * P->len_fq;
  P->len2 = 4 * P->len_fq;
  P->lenT = 12 * P->len_fq;
  set_zr(P, r);
  // SURVEY.md 8d (instrumented reference, f.param: 158-bit r, 472-bit tateexp): 54 k F_q products in the
  // Miller loop, 118 k in f_tateexp; other sizes scale with the two loop lengths
  P->fq_muls_single = 54000.0 * r.bits() / 158.0 + 118887.0 * te.bits() / 472.0;
  P->fq_muls_prod_a = P->fq_muls_single;   // generic_prod_pairings: k full pairings
  P->fq_muls_prod_b = 0.0;
  if (fill_ext_sqrt(P, q, 2)) return 1;    // G2 = E'(F_q^2)
  return fill_hash_consts(P, q, nullptr);  // no cofactor (f_param.c:372)
}


// E(F_q) coefficients of the pairing's G1 curve for the group-operation kernels (Montgomery words)
static void fill_curve(const pbc_hip_pairing_s *P, CurveK &C) {
  memset(&C, 0, sizeof C);
  if ((P->type == 'a' || P->type == '1') && P->nlimb == 33) {   // y^2 = x^3 + x (a_param.c:1450-1452, :2247-2251)
    memcpy(C.a, P->k33.one, sizeof P->k33.one);
  } else if (P->type == 'a' || P->type == '1') {
    memcpy(C.a, P->k16.one, sizeof P->k16.one);
  } else if (P->type == 'e') {
    memcpy(C.a, P->econst.A, sizeof P->econst.A);
    memcpy(C.b, P->econst.B, sizeof P->econst.B);
  } else if (P->type == 'd' || P->type == 'g') {
    memcpy(C.a, P->dconst.A, sizeof P->dconst.A);
    memcpy(C.b, P->dconst.B, sizeof P->dconst.B);
  } else {                              // y^2 = x^3 + b (f_param.c:365-367)
    memcpy(C.b, P->fconst.B, sizeof P->fconst.B);
    C.a_is_zero = 1;
  }
  memcpy(C.cofac, P->hash.cofac, sizeof C.cofac);
  C.cofbits = P->hash.cofbits;
  C.sqrt_mode = P->hash.sqrt_mode;
  memcpy(C.sqrt_e, P->hash.sqrt_e, sizeof C.sqrt_e);
  C.sqrt_bits = P->hash.sqrt_bits;
  C.ts_s = P->hash.ts_s;
  memcpy(C.ts_c, P->hash.ts_c, sizeof C.ts_c);
}

// The constant block of a pairing object as the kernels receive it, as their last argument (layout: fp.cuh, "KArgs").  Built on the host for
// every launch from the object's own copies -- nothing lives in device globals.
template <int N> static const FpK<N> &host_fpk(const pbc_hip_pairing_s *P);
#define PBC_HOST_FPK_OF(n) template <> const FpK<n> &host_fpk<n>(const pbc_hip_pairing_s *P) { return P->k##n; }
PBC_FOR_EACH_N(PBC_HOST_FPK_OF)
#undef PBC_HOST_FPK_OF
// for_pairing: the block of a pairing / product launch (type f: the i-basis constants when the object has them)
template <int N>
static void fill_kargs(const pbc_hip_pairing_s *P, KArgs<N> &K, bool for_pairing = false) {
  memset(&K, 0, sizeof K);
  CurveK C;
  fill_curve(P, C);
  memcpy(K.head + KOFF_CURVE, &C, sizeof C);
  if (P->type == 'a' || P->type == '1') memcpy(K.head + KOFF_TYPE, &P->a, sizeof P->a);
  else if (P->type == 'd' || P->type == 'g') memcpy(K.head + KOFF_TYPE, &P->dconst, sizeof P->dconst);
  else if (P->type == 'f') memcpy(K.head + KOFF_TYPE, for_pairing && P->f_bm1 ? &P->fconst_i : &P->fconst, sizeof P->fconst);
  else if (P->type == 'e') memcpy(K.head + KOFF_TYPE, &P->econst, sizeof P->econst);
  memcpy(K.head + KOFF_XS, &P->xs, sizeof P->xs);
  const uint32_t opt = P->no_fair ? 1u : 0u;
  memcpy(K.head + KOFF_OPT, &opt, 4);
  K.fp = host_fpk<N>(P);
}

// the constant block of the Z_r arithmetic: FpK for the modulus r in the smallest built-in width that holds it (set_zr);
// the kernels of pbc_hip_group2.hip read nothing but the FpK part
template <int N>
static int zr_kargs(const pbc_hip_pairing_s *P, KArgs<N> &K) {
  memset(&K, 0, sizeof K);
  pbc_host::Big r;
  r.w.assign(P->zr_words, P->zr_words + 34);
  r.trim();
  if (fill_fpk<N>(K.fp, r, 2)) return fail("Z_r: the group order does not fit the %d-word arithmetic", N);
  return 0;
}
