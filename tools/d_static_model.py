#!/usr/bin/env python3
"""Static model of the dynamic VALU instruction count of one type d (d159) pairing, from the gfx950 assembly
`make -C pbc_amd` leaves in /tmp/pbc_hip_build: every out-of-line routine is straight-line code, so

    instructions per pairing = sum over loop bodies of (inline instructions + sum of callee sizes) x trip count

with the trip counts fixed by the parameter file (Miller loop over r, Lucas ladder over Phi_6(q)/r).  For the
default kernel the model gives 2.60 M against 2.61 M measured with the SQ_INSTS_VALU counter
(profiles/r01_more_pmc_dfg.json), which is what makes it usable as a GPU-less price tag for variants of the
kernel (profiles/r01_notes.md).  Block selection inside the kernel body is heuristic: the block calling the
tangent routine holds the inline code of both F_q^6 products of a Miller step, the next calling block that
of the F_q^6 square, the two small blocks calling the F_q^3 square that of a ladder step.

  tools/d_static_model.py [FILE.s] [PARAM]
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_count  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def valu(blocks):
    t = isa_count.totals(blocks)
    return t.get("valu", 0) + t.get("mad64", 0)


def calls_of(block):
    out = []
    for _, s in block["ins"]:
        m = re.search(r"(_Z\w+)@rel32@lo", s)
        if m and "_ZN3pbc" in m.group(1) and "c_d" not in m.group(1)[:12]:
            out.append(m.group(1))
    return out


def model(funcs, kernel_pat, fn_pat, names, trips):
    size = {}
    for short, pat in names.items():
        hit = [n for n in funcs if re.search(fn_pat, n) and re.search(pat, n)]
        if not hit:
            raise SystemExit(f"no routine matches {pat}")
        size[short] = valu(funcs[hit[0]])
    kern = [n for n in funcs if re.search(kernel_pat, n)][0]
    blocks = list(funcs[kern].items())
    main = next(i for i, (_, b) in enumerate(blocks) if any("dbl_line_fn" in c for c in calls_of(b)))
    sqr = next(i for i in range(main + 1, len(blocks)) if calls_of(blocks[i][1]))
    small = [i for i, (bn, b) in enumerate(blocks) if i > sqr and any("f3_sqr_call" in c for c in calls_of(b)) and valu({bn: b}) < 200]
    inl_mul = valu(dict([blocks[main]])) / 2
    inl_sqr = valu(dict([blocks[sqr]]))
    inl_luc = sum(valu(dict([blocks[i]])) for i in small[:2])
    line_extra = size.get("line_y", 0)
    f6mul = inl_mul + 3 * size["f3_mul"] + size["mul_v"]
    per = {
        "tangent step": size["dbl_line"] + line_extra + f6mul,
        "chord step": size["add_line"] + line_extra + f6mul,
        "F_q^6 square": inl_sqr + 2 * size["f3_mul"] + 2 * size["mul_v"],
        "ladder step": inl_luc + size["f3_mul"] + size["f3_sqr"],
    }
    total = sum(per[k] * trips[k] for k in per)
    rest = valu(funcs[kern]) - valu(dict([blocks[main]])) - inl_sqr - inl_luc + 40 * size["f3_mul"]
    return size, per, total + rest, rest


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pbc_hip_build/obj_libpbc_hip/pbc_hip_d-hip-amdgcn-amd-amdhsa-gfx950.s"
    param = sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "..", "pbc_amd", "param", "d159.param")
    kv = dict(re.findall(r"(\w+)\s+(\S+)", open(param).read()))
    q, r = int(kv["q"]), int(kv["r"])
    phik = (q * q - q + 1) // r
    rb = r.bit_length()
    trips = {"tangent step": rb - 1, "chord step": bin(r).count("1") - 2, "F_q^6 square": rb - 2, "ladder step": phik.bit_length()}
    funcs = isa_count.parse(path)
    variants = [
        ("default (saturated words)", r"^_Z21d_prod_pairing_kernelILi5ELi3", r"TypeMNTILi5ELi3E",
         {"f3_mul": "f3_mul_call", "f3_sqr": "f3_sqr_call", "mul_v": "f3_mul_v_call", "dbl_line": "d_dbl_line_fn", "add_line": "d_add_line_fn"}),
    ]
    base = None
    for title, kpat, fpat, names in variants:
        if not any(re.search(kpat, n) for n in funcs):
            continue
        size, per, total, rest = model(funcs, kpat, fpat, names, trips)
        print(title)
        print("  routine sizes (VALU instructions):", ", ".join(f"{k} {v}" for k, v in size.items()))
        for k in per:
            print(f"  {k:14s} {per[k]:8.0f} x {trips[k]:4d} = {per[k] * trips[k] / 1e6:6.3f} M")
        print(f"  outside the loops ~{rest / 1e3:.0f} k;  total {total / 1e6:.3f} M VALU instructions per pairing"
              + (f"  ({100 * (total / base - 1):+.1f} %)" if base else ""))
        base = base or total


if __name__ == "__main__":
    main()
