// dw_sched.h -- HOST side of the wave-per-pairing type d kernel (pairing_dw.cuh): one pairing flattened into a SCHEDULE of packed
// entries.  Which level of which program runs when depends on the curve's constants only (the signed digits of r, the bits
// of Phi_6(q) / r), so the host walks the two tracks of the Miller loop and the final exponentiation ONCE per object and the
// kernel is a plain interpreter: entry -> rows -> sums (no scheduling arithmetic, no table lookups on the device; the first
// version, with the state machine in the kernel, spent ~100 scalar instructions and 5 scalar loads per level on it).
// tools/dw_gen.py holds the same walk on Python integers (Model.pairing / flat_schedule); tests compare the two.
//   entry (64 bits): row a [0:12) | lanes a [12:17) | row b [17:29) | lanes b [29:34) | terms [34:38) | op [38:42) |
//                    table line [42:54) | line flag [55]
//   op 0: a level (track a on the first lanes, track b on the last);  1: the B = 0 test of cc_tatepower (lane code);
//   2: the inversion (lane code);  3: end;  4: table line -> coefficient bank (pairing_pp_apply, before the first level)
//   line flag on a level: that line of the pairing_pp table arrives in coefficient bank (line % 2) when the level is over
// Four schedules per object, one after the other in one buffer (Sched::off): the pairing; the Miller value alone (a TERM of a
// product); the product with a term's value (two levels, the kernel repeats them) + the final exponentiation; pairing_pp_apply.
#pragma once
#include <stdint.h>
#include <vector>
#include "dw_tables.h"

namespace pbc { namespace dw {

inline uint64_t entry(int a, int b, int op = OP_LEVEL, int line = -1) {   // a, b: indices into h_level, -1: the track idles
  const LevelRef z = {0, 0, 0};
  const LevelRef &A = a >= 0 ? h_level[a] : z, &B = b >= 0 ? h_level[b] : z;
  const unsigned T = (A.lanes ? A.T : 0u) > (B.lanes ? B.T : 0u) ? A.T : (B.lanes ? B.T : 0u);
  return (uint64_t) A.row | (uint64_t) A.lanes << 12 | (uint64_t) B.row << 17 | (uint64_t) B.lanes << 29 | (uint64_t) T << 34 | (uint64_t) op << 38 |
         (line >= 0 ? (uint64_t) line << 42 | 1ull << 55 : 0ull);
}
constexpr int kMaxLines = 4096;                                   // (the entry's line field)
inline void run_levels(std::vector<uint64_t> &out, int first, int count) { for (int i = 0; i < count; i++) out.push_back(entry(first + i, -1)); }
// cc_tatepower with one inversion (pairing_d.cuh d_final_exp), then the end marker
inline void build_final(std::vector<uint64_t> &out, const uint32_t *phik, int phikbits) {
  run_levels(out, P_fe1, N_fe1);
  out.push_back(entry(-1, -1, OP_BZERO));
  run_levels(out, P_fe2, N_fe2);
  out.push_back(entry(-1, -1, OP_INV));
  run_levels(out, P_fe3, N_fe3);
  for (int j = phikbits - 1; j >= 0; j--) {                       // lucas_even (d_param.c:462-482): j == 0 takes the 0-branch
    const bool bit = j ? ((phik[j >> 5] >> (j & 31)) & 1) != 0 : false;
    run_levels(out, bit ? P_lucas1 : P_lucas0, N_lucas0);
  }
  run_levels(out, P_fe4, N_fe4);
  out.push_back(entry(-1, -1, OP_END));
}
template <class Digit>                                            // Digit(m): the signed digit of the Miller loop at position m
inline void build_miller(std::vector<uint64_t> &out, int rbits, const Digit &digit) {
  // The accumulator track: for m = rbits - 2 .. 0: product with the tangent's line; product with the chord's if the digit is
  // set (m > 0); square (m > 0) -- two levels each.  The point track: the same steps without the squares, ahead of it:
  //   * point program j leaves the COEFFICIENTS of line j in bank j % 2 and, in its first two levels, evaluates line j - 1
  //     (the other coefficient bank) into value bank (j - 1) % 2; after the last one a short program evaluates the last line;
  //   * the product with line i (both levels read value bank i % 2) starts when point program i + 1 has run two levels;
  //   * point program j starts when the product with line j - 3 is COMPLETE (its first levels overwrite that value bank).
  int fm = rbits - 2, pm = rbits - 2, fph = 0, pph = 0, lines_taken = 0, line_ready = 0, pstarted = 0, fmul_done = 0;
  int fbase = -1, flev = 0, fcount = 0, pbase = -1, plev = 0, pcount = 0;
  bool f_is_mul = false, p_exhausted = false;             // p_exhausted: the closing evaluation has been started
  for (;;) {
    if (fbase < 0) {
      while (fm >= 0 && ((fph == 1 && !(fm > 0 && digit(fm))) || (fph == 2 && fm <= 0))) { if (++fph == 3) { fph = 0; fm--; } }
      if (fm < 0) { if (pbase < 0) break; }
      else if (fph == 2) { fbase = P_f_sqr; fcount = N_f_sqr; flev = 0; f_is_mul = false; fph = 0; fm--; }
      else if (lines_taken < line_ready) {
        fbase = (lines_taken & 1) ? P_f_mul1 : P_f_mul0; fcount = N_f_mul0; flev = 0; f_is_mul = true;
        lines_taken++;
        fph++;
      }
    }
    if (pbase < 0 && !p_exhausted) {
      while (pm >= 0 && pph == 1 && !(pm > 0 && digit(pm))) { pph = 0; pm--; }
      if (pstarted < 3 || fmul_done >= pstarted - 2) {
        const int bank = pstarted & 1;
        if (pm < 0) { pbase = bank ? P_pt_eval1 : P_pt_eval0; pcount = N_pt_eval0; p_exhausted = true; }
        else if (pph == 0) { pbase = bank ? P_pt_dbl1 : P_pt_dbl0; pcount = N_pt_dbl0; pph = 1; }
        else {
          const bool neg = digit(pm) < 0;
          pbase = neg ? (bank ? P_pt_addm1 : P_pt_addm0) : (bank ? P_pt_addp1 : P_pt_addp0); pcount = N_pt_addp0;
          pph = 0; pm--;
        }
        plev = 0;
        pstarted++;
      }
    }
    out.push_back(entry(fbase >= 0 ? fbase + flev : -1, pbase >= 0 ? pbase + plev : -1));
    if (fbase >= 0 && ++flev == fcount) { fbase = -1; fmul_done += f_is_mul ? 1 : 0; }
    if (pbase >= 0) {
      ++plev;
      if (plev == 2 && pstarted >= 2) line_ready = pstarted - 1;
      if (plev == pcount) pbase = -1;
    }
  }
}
// pairing_pp_apply: the accumulator track as above; the other track only EVALUATES -- line i (coefficient bank i % 2, loaded beside
// the first level of the evaluation of line i - 1) -> value bank i % 2, two levels;
//   * the product with line i starts when its evaluation is complete;
//   * the evaluation of line i starts when the product with line i - 2 is complete (it overwrites that value bank).
template <class Digit>
inline void build_pp_miller(std::vector<uint64_t> &out, int rbits, const Digit &digit) {
  int nl = 0;
  for (int m = rbits - 2; m >= 0; m--) nl += 1 + ((m > 0 && digit(m)) ? 1 : 0);
  out.push_back(entry(-1, -1, OP_LOADLINE, 0));
  int fm = rbits - 2, fph = 0, lines_taken = 0, evals_done = 0, fmul_done = 0, pi = 0;
  int fbase = -1, flev = 0, fcount = 0, pbase = -1, plev = 0;
  bool f_is_mul = false;
  for (;;) {
    if (fbase < 0) {
      while (fm >= 0 && ((fph == 1 && !(fm > 0 && digit(fm))) || (fph == 2 && fm <= 0))) { if (++fph == 3) { fph = 0; fm--; } }
      if (fm < 0) { if (pbase < 0) break; }
      else if (fph == 2) { fbase = P_f_sqr; fcount = N_f_sqr; flev = 0; f_is_mul = false; fph = 0; fm--; }
      else if (lines_taken < evals_done) {
        fbase = (lines_taken & 1) ? P_f_mul1 : P_f_mul0; fcount = N_f_mul0; flev = 0; f_is_mul = true;
        lines_taken++;
        fph++;
      }
    }
    if (pbase < 0 && pi < nl && (pi < 2 || fmul_done >= pi - 1)) {
      pbase = ((pi + 1) & 1) ? P_pt_eval1 : P_pt_eval0;           // pt_eval{b} evaluates the OTHER bank
      plev = 0;
      pi++;
    }
    const int line = (pbase >= 0 && plev == 0 && pi < nl) ? pi : -1;
    out.push_back(entry(fbase >= 0 ? fbase + flev : -1, pbase >= 0 ? pbase + plev : -1, OP_LEVEL, line));
    if (fbase >= 0 && ++flev == fcount) { fbase = -1; fmul_done += f_is_mul ? 1 : 0; }
    if (pbase >= 0 && ++plev == N_pt_eval0) { pbase = -1; evals_done++; }
  }
}
enum { SCHED_PAIRING = 0, SCHED_MILLER = 1, SCHED_FINISH = 2, SCHED_PP = 3 };
// Sched (host_params.h DwSched): e -- the four schedules one after the other; off[4] -- where each begins; lines -- the lines of a
// pairing_pp table
template <class Sched, class Digit>
inline void build_schedules(Sched &S, int rbits, const Digit &digit, const uint32_t *phik, int phikbits) {
  S.e.clear();
  S.lines = 0;
  for (int m = rbits - 2; m >= 0; m--) S.lines += 1 + ((m > 0 && digit(m)) ? 1 : 0);
  S.off[SCHED_PAIRING] = S.e.size();
  build_miller(S.e, rbits, digit);
  build_final(S.e, phik, phikbits);
  S.off[SCHED_MILLER] = S.e.size();
  build_miller(S.e, rbits, digit);
  S.e.push_back(entry(-1, -1, OP_END));
  S.off[SCHED_FINISH] = S.e.size();
  run_levels(S.e, P_f_mul0, N_f_mul0);
  build_final(S.e, phik, phikbits);
  S.off[SCHED_PP] = S.e.size();
  build_pp_miller(S.e, rbits, digit);
  build_final(S.e, phik, phikbits);
}

} }  // namespace pbc::dw
