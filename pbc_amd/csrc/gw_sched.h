// gw_sched.h -- HOST side of the wave-per-pairing type g kernel (pairing_gw.cuh): one pairing as a straight line of packed schedule
// entries for the machine of pairing_dw.cuh (dw_sched.h's entry format; one track).  The order of the programs depends on the curve's
// constants only -- the signed digits of r (cc_miller_no_denom_affine, the loop of ecc/d_param.c:321-422 that ecc/g_param.c shares) and
// the bits of Phi_10(q) / r (lucas_even of cc_tatepower, g_param.c:471-558) -- so the host writes it once per object.
// tools/gw_gen.py holds the same sequence on Python integers (sequence / flat_schedule); tests compare.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "gw_tables.h"

namespace pbc { namespace gw {

struct Builder {
  std::vector<uint64_t> &out;
  bool ok = true;
  explicit Builder(std::vector<uint64_t> &o) : out(o) {}
  void run(const std::string &name) {                               // every level of the program of that name
    for (int i = 0; i < kProgs; i++)
      if (name == h_prog[i].name) {
        for (int l = 0; l < h_prog[i].count; l++) {
          const LevelRef &A = h_level[h_prog[i].first + l];
          out.push_back((uint64_t) A.row | (uint64_t) A.lanes << 12 | (uint64_t) A.T << 34 | (uint64_t) OP_LEVEL << 38);
        }
        return;
      }
    ok = false;
  }
  void op(int o, int line = -1) { out.push_back((uint64_t) o << 38 | (line >= 0 ? (uint64_t) line << 42 | 1ull << 55 : 0ull)); }
  void run_with_line(const std::string &name, int line) {          // ... whose first level also takes table line `line` (-1: none)
    const size_t at = out.size();
    run(name);
    if (line >= 0 && out.size() > at) out[at] |= (uint64_t) line << 42 | 1ull << 55;
  }
};
template <class Digit>                                              // digit(m): the signed digit of the Miller loop at position m
inline void build_miller(Builder &B, int rbits, const Digit &digit) {
  // (a square and the doubling of the step after it share nothing: ONE program, "sqrdbl")
  for (int m = rbits - 2; m >= 0; m--) {
    if (m == rbits - 2) B.run("pt_dbl");
    B.run("line_mul");
    if (m > 0 && digit(m)) { B.run(digit(m) < 0 ? "pt_addm" : "pt_addp"); B.run("line_mul"); }
    if (m > 0) B.run("sqrdbl");
  }
}
inline void build_final(Builder &B, const uint32_t *phik, int phikbits) {
  // cc_tatepower with one inversion (pairing_d.cuh d_final_exp)
  B.run("fe1"); B.op(OP_BZERO); B.run("fe2"); B.op(OP_INV); B.run("fe3");
  for (int j = phikbits - 1; j >= 0; j--) {                         // lucas_even: j == 0 takes the 0-branch
    const bool bit = j ? ((phik[j >> 5] >> (j & 31)) & 1) != 0 : false;
    B.run(bit ? "lucas1" : "lucas0");
  }
  B.run("fe4");
  B.op(OP_END);
}
enum { SCHED_PAIRING = 0, SCHED_MILLER = 1, SCHED_FINISH = 2, SCHED_PP = 3 };
constexpr int kMaxLines = 4096;
// Sched (host_params.h DwSched): the pairing; the Miller value alone (a TERM of a product); the product with a term's value (mul_f_u, a
// mark: the kernel repeats the entries before it) + the final exponentiation; pairing_pp_apply (line i of the table -> La, Lb, Lc: the
// first by an entry of its own, line i + 1 beside the first level of the product with line i)
template <class Sched, class Digit>
inline bool build_schedules(Sched &S, int rbits, const Digit &digit, const uint32_t *phik, int phikbits) {
  S.e.clear();
  Builder B(S.e);
  S.off[SCHED_PAIRING] = S.e.size();
  build_miller(B, rbits, digit);
  build_final(B, phik, phikbits);
  S.off[SCHED_MILLER] = S.e.size();
  build_miller(B, rbits, digit);
  B.op(OP_END);
  S.off[SCHED_FINISH] = S.e.size();
  B.run("mul_f_u");
  B.op(OP_MARK);
  build_final(B, phik, phikbits);
  S.off[SCHED_PP] = S.e.size();
  S.lines = 0;
  for (int m = rbits - 2; m >= 0; m--) S.lines += 1 + ((m > 0 && digit(m)) ? 1 : 0);
  B.op(OP_LOADLINE, 0);
  int i = 0;
  for (int m = rbits - 2; m >= 0; m--) {
    const int nl = 1 + ((m > 0 && digit(m)) ? 1 : 0);
    for (int t = 0; t < nl; t++) {
      B.run("ln_eval");
      B.run_with_line("line_mul", i + 1 < S.lines ? i + 1 : -1);
      i++;
    }
    if (m > 0) B.run("f_sqr");
  }
  build_final(B, phik, phikbits);
  return B.ok;
}

} }  // namespace pbc::gw
