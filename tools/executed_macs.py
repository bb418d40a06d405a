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
    path = os.path.join(ROOT, "profiles", "executed_macs.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
