#!/usr/bin/env python3
"""Multiply-adds the KERNEL SOURCE executes per unit of every bench workload, counted by running one unit through the
host-compiled mirror of pbc_amd/csrc/*.cuh (tests/hostsim; every multiplier body of fp.cuh reports its 32 x 32 + 64-bit
multiply-adds through the PBC_COUNT_MACS hook).  Control flow is data-independent, so one unit is exact for all.
Writes profiles/executed_macs.json, which bench.py reads for roofline.executed_macs_per_unit -- the work the kernel
does, as opposed to algorithmic_macs_per_unit, the work the REFERENCE's algorithm would do.

    python tools/executed_macs.py            # CPU only, about a minute
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402  (WORKLOADS table; bench.py imports torch, which is fine on CPU)
import hostsim  # noqa: E402


def main():
    out = {}
    for name, (pname, fixture, k, _, desc) in sorted(bench.WORKLOADS.items()):
        S = hostsim.HostSim(open(os.path.join(ROOT, "pbc_amd", "param", pname + ".param")).read())
        g1, g2, _ = bench.load_vec(os.path.join(ROOT, "tests", "golden", fixture))
        reps = -(-k // g1.shape[0])
        g1, g2 = np.tile(g1, (reps, 1))[:k], np.tile(g2, (max(reps, 2), 1))[:max(k, 2)]
        if name.endswith("-pp"):
            S.pp(g1[0], g2[:1])                      # init + one apply ...
            a = S.macs(reset=True)
            S.pp(g1[0], g2[:2])                      # ... init + two applies: the difference is one apply
            macs = S.macs(reset=True) - a
        else:
            S.macs(reset=True)
            S.prod_pairing(g1, g2[:k], k)
            macs = S.macs(reset=True)
        out[name] = {"executed_macs_per_unit": macs, "workload": desc, "terms_per_unit": k}
        print("%-9s %12d multiply-adds per unit   (%s)" % (name, macs, desc))
    # the group operations (bench_group.py): one unit through the routines the library's fast pass runs.  Scalar
    # multiplications, powers and table look-ups are data-independent; element_from_hash retries x <- x^2 + 1 until
    # x^3 + a x + b is a square (two tries on average per LANE; a wave repeats until its slowest lane is done), so its
    # count is the mean over 64 digests of one lane's tries -- the work the wave executes beyond that is not counted.
    import bench_group
    rng = np.random.default_rng(99)
    for name, (pname, fixture, op, _) in sorted(bench_group.GROUP_WORKLOADS.items()):
        text = open(os.path.join(ROOT, "pbc_amd", "param", pname + ".param")).read()
        S = hostsim.HostSim(text)
        g1, g2, gt = bench.load_vec(os.path.join(ROOT, "tests", "golden", fixture))
        zl = (bench_group.param_int(text, "r").bit_length() + 7) // 8
        z = rng.integers(0, 256, (1, zl), dtype=np.uint8)
        z[0, 0] &= 0x3f
        S.fallbacks()
        S.macs(reset=True)
        if op == "g1mul":
            S.group(0, g1[:1], z)
        elif op == "g2mul":
            S.g2_mul(g2[:1], z)
        elif op == "gtpow":
            S.group(2, gt[:1], z)
        elif op == "hashg1":
            S.from_hash(rng.integers(0, 256, (64, 32), dtype=np.uint8), 32)
        elif op in ("compress", "decompress"):
            comp = S.compress(0, g1[:1])
            S.macs(reset=True)
            if op == "compress":
                S.compress(0, g1[:1])
            else:
                S.compress(1, comp)
        elif op == "g1add":
            S.affine_op(0, 1, g1[:1], g1[1:2])
        elif op == "zrinv":
            z[0, -1] |= 1
            S.zr_op(3, z)
        elif op in ("g1pow2", "gtpow2"):
            # the default route: two single-base ladders / powers and one addition / product (pbc_hip_group2.hip multi_launch)
            z2 = z[:, ::-1].copy()
            z2[0, 0] &= 0x3f
            if op == "g1pow2":
                S.group(0, g1[:1], z)
                S.group(0, g1[1:2], z2)
                S.affine_op(0, 1, g1[:1], g1[1:2])
            else:
                S.group(2, gt[:1], z)
                S.group(2, gt[1:2], z2)
                S.group(1, gt[:1], gt[1:2])
        elif op in ("g1pp", "gtpp"):
            grp = 1 if op == "g1pp" else 3
            base = g1[5] if grp == 1 else gt[5]
            S.element_pp(grp, base, z, zl)                   # table + one power ...
            a = S.macs(reset=True)
            S.element_pp(grp, base, np.concatenate([z, z]), zl)   # ... table + two powers
        else:
            continue                                         # bls-verify: the sum of its parts, below
        macs = S.macs(reset=True)
        if op in ("g1pp", "gtpp"):
            macs -= a
        if op == "hashg1":
            macs //= 64
        assert S.fallbacks() == 0, name
        out[name] = {"executed_macs_per_unit": macs, "workload": "%s (%s.param)" % (bench_group.DESC[op], pname)}
        print("%-12s %12d multiply-adds per unit   (%s, %s.param)" % (name, macs, bench_group.DESC[op], pname))
    # one signature of a-bls-verify: a hash, two scalar multiplications, two terms of a 16-term product
    out["a-bls-verify"] = {"executed_macs_per_unit": out["a-hash-g1"]["executed_macs_per_unit"] + 2 * out["a-g1-mul"]["executed_macs_per_unit"]
                           + out["a-prod16"]["executed_macs_per_unit"] // 8,
                           "workload": "BLS batch verification (a.param): hash + 2 x mul_zn + 2 of the 16 terms of a product"}
    path = os.path.join(ROOT, "profiles", "executed_macs.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
