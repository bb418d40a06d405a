"""Instruction-throughput probes on the GPU (feeds DESIGN.md's roofline section)."""
import json, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.makedirs("gpurun_out", exist_ok=True)
import pbc_amd
names = {0: "v_mad_u64_u32 (8 chains)", 1: "mad+addc MAC (1 chain)", 2: "v_mul_lo_u32", 3: "v_mul_hi_u32",
         4: "v_dot2_u32_u16", 5: "v_fma_f64", 6: "v_addc_co_u32 chain", 7: "v_mad_u32_u24",
         8: "v_lshl_add_u64", 9: "v_dot4_u32_u8", 10: "mad+addc+s_nop", 11: "v_mad_u32_u16",
         12: "v_add_u32", 13: "v_mad_u64_u32 sgpr operand", 14: "v_mad_u64_u32 sgpr operand, 4 carry-out pairs",
         15: "v_mad_i64_i32 sgpr operand", 16: "v_mad_u64_u32 VGPR factors in the two free banks",
         17: "v_mad_u64_u32 VGPR factors in the accumulator's banks", 18: "v_mad_u64_u32 VGPR factors in one free bank"}
res = {}
for v, nm in names.items():
    r, ms = pbc_amd.int_mac_peak(v, 3000)
    # cycles per wave-instruction per SIMD at 2.4 GHz: 256 CU * 4 SIMD * 2.4e9 / (lane-ops/64)
    cyc = 256 * 4 * 2.4e9 / (r / 64.0)
    res[nm] = {"lane_ops_per_s": r, "ms": ms, "cycles_per_wave_instr_at_2.4GHz": cyc}
    print("%-28s %8.3f T lane-ops/s  %7.2f cyc/wave-instr/SIMD  (%.2f ms)" % (nm, r / 1e12, cyc, ms))
json.dump(res, open("gpurun_out/probe.json", "w"), indent=1)

mnames = {1: "mul 29-bit unsat", 2: "mul 29-bit unsat, 2 acc",
          3: "sqr 29-bit", 4: "sqr 29-bit, 2 acc", 5: "safegcd inversion (+1 add)", 6: "mul 29-bit out-of-line call",
          7: "sqr 29-bit out-of-line call"}
mres = {}
for v, nm in mnames.items():
    for wps in (1, 2, 4):
        r, ms = pbc_amd.mul_bench(v, 40 if v == 5 else 300, wps)
        cyc = 256 * 4 * 2.4e9 / (r / 64.0)
        mres["%s @%dw" % (nm, wps)] = {"fq_mul_per_s": r, "ms": ms}
        print("%-32s %d waves/SIMD: %8.3f G mul/s  %8.0f cyc/wave-mul/SIMD @2.4GHz" % (nm, wps, r / 1e9, cyc))
json.dump(mres, open("gpurun_out/mulbench.json", "w"), indent=1)
