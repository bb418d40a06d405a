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
  void op(int o) { out.push_back((uint64_t) o << 38); }
};
template <class Digit>                                              // digit(m): the signed digit of the Miller loop at position m
inline bool build_schedule(std::vector<uint64_t> &out, int rbits, const Digit &digit, const uint32_t *phik, int phikbits) {
  out.clear();
  Builder B(out);
  // (a square and the doubling of the step after it share nothing: ONE program, "sqrdbl")
  for (int m = rbits - 2; m >= 0; m--) {
    if (m == rbits - 2) B.run("pt_dbl");
    B.run("line_mul");
    if (m > 0 && digit(m)) { B.run(digit(m) < 0 ? "pt_addm" : "pt_addp"); B.run("line_mul"); }
    if (m > 0) B.run("sqrdbl");
  }
  // cc_tatepower with one inversion (pairing_d.cuh d_final_exp)
  B.run("fe1"); B.op(OP_BZERO); B.run("fe2"); B.op(OP_INV); B.run("fe3");
  for (int j = phikbits - 1; j >= 0; j--) {                         // lucas_even: j == 0 takes the 0-branch
    const bool bit = j ? ((phik[j >> 5] >> (j & 31)) & 1) != 0 : false;
    B.run(bit ? "lucas1" : "lucas0");
  }
  B.run("fe4");
  B.op(OP_END);
  return B.ok;
}

} }  // namespace pbc::gw
