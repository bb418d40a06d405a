// fw_sched.h -- HOST side of the wave-per-pairing type f kernel (pairing_fw.cuh): one pairing as a straight line of packed schedule
// entries for the machine of pairing_dw.cuh (dw_sched.h's entry format; one track).  The order of the programs depends on the curve's
// constants only -- the signed digits of r (cc_miller_no_denom, ecc/f_param.c:216-233) and the bits and the sign of the BN parameter
// x (the hard part of f_tateexp, :250-283, as pairing_f.cuh f_hard_bn runs it) -- so the host writes it once per object.
// tools/fw_gen.py holds the same sequence on Python integers (miller_sequence / final_sequence / flat_schedule); tests compare.
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include "fw_tables.h"

namespace pbc { namespace fw {

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
    ok = false;                                                     // (a program the generator did not emit)
  }
  void op(int o) { out.push_back((uint64_t) o << 38); }
  // dst <- a^|x| by square-and-multiply on dst, conjugated for a negative x (f12_pow_x)
  void pow_x(const char *dst, const char *a, uint64_t x, bool xneg) {
    const std::string d(dst), s(a);
    run("copy_" + d + "_" + s);
    int top = 63;
    while (top > 0 && !((x >> top) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
      run("sqr_" + d + "_" + d);
      if ((x >> i) & 1) run("mul_" + d + "_" + d + "_" + s);
    }
    if (xneg) run("conj_" + d + "_" + d);
  }
};
// Miller: digit(m) is the signed digit of position m
template <class Digit>
inline void build_miller(Builder &B, int rbits, const Digit &digit) {
  // (a square and the doubling of the step after it share nothing: ONE program, "sqrdbl", whose four levels hold the square's two)
  for (int m = rbits - 2; m >= 0; m--) {
    if (m == rbits - 2) B.run("pt_dbl");
    B.run("line_mul");
    if (m > 0 && digit(m)) { B.run(digit(m) < 0 ? "pt_addm" : "pt_addp"); B.run("line_mul"); }
    if (m > 0) B.run("sqrdbl");
  }
}
inline void build_final(Builder &B, uint64_t x, bool xneg) {
  // easy part: F <- F^(q^8) F^(q^6) / (F^(q^2) F), the inverse by polymod_invert's norm trick (poly.c:521-536)
  for (const char *p : {"qp2_Y_F", "qp2_Y_Y", "qp2_Y_Y", "qp2_Y_Y", "conj_U_F", "mul_Y_Y_U", "qp2_U_F", "mul_U_U_F", "qp2_T1_U", "copy_T0_T1"}) B.run(p);
  for (int i = 0; i < 4; i++) { B.run("qp2_T1_T1"); B.run("mul_T0_T0_T1"); }
  B.run("mul_T1_U_T0"); B.run("norm_T1"); B.op(OP_INV); B.run("scinv_U_T0_T1");
  B.run("mul_F_Y_U");
  // hard part: the BN vector chain (pairing_f.cuh f_hard_bn)
  B.pow_x("FX", "F", x, xneg); B.pow_x("FX2", "FX", x, xneg); B.pow_x("FX3", "FX2", x, xneg);
  for (const char *p : {"frob_U_FX3", "mul_Y_U_FX3", "conj_Y_Y", "sqr_T0_Y",
                        "frob_U_FX2", "mul_Y_U_FX", "conj_Y_Y", "mul_T0_T0_Y",
                        "conj_Y_FX2", "mul_T0_T0_Y",
                        "frob_U_FX", "conj_U_U", "mul_T1_U_Y", "mul_T1_T1_T0",
                        "qp2_Y_FX2", "mul_T0_T0_Y", "sqr_T1_T1", "mul_T1_T1_T0", "sqr_T1_T1",
                        "conj_Y_F", "mul_T0_T1_Y",
                        "frob_U_F", "qp2_Y_F", "mul_FX_U_Y", "frob_U_Y", "mul_FX_FX_U", "mul_T1_T1_FX",
                        "sqr_T0_T0", "mul_F_T0_T1"}) B.run(p);
  B.op(OP_END);
}
enum { SCHED_PAIRING = 0, SCHED_MILLER = 1, SCHED_FINISH = 2 };
// Sched (host_params.h DwSched): e -- the schedules one after the other; off[] -- where each begins: the pairing; the Miller value
// alone (a TERM of a product); the product with a term's value (the three levels of mul_F_F_U, the kernel repeats them) + the final
// exponentiation
template <class Sched, class Digit>
inline bool build_schedules(Sched &S, int rbits, const Digit &digit, uint64_t x, bool xneg) {
  S.e.clear();
  Builder B(S.e);
  S.off[SCHED_PAIRING] = S.e.size();
  build_miller(B, rbits, digit);
  build_final(B, x, xneg);
  S.off[SCHED_MILLER] = S.e.size();
  build_miller(B, rbits, digit);
  B.op(OP_END);
  S.off[SCHED_FINISH] = S.e.size();
  B.run("mul_F_F_U");
  build_final(B, x, xneg);
  S.off[3] = S.e.size();
  return B.ok;
}

} }  // namespace pbc::fw
