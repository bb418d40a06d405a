#!/usr/bin/env python3
"""The reference's CPU rate for EVERY bench workload, measured in the build container (the reference needs no GPU):
the same `cpu_baseline` legs bench.py / bench_group.py run on the GPU box for the headline workloads, here for all of
them, written to profiles/r05_cpu_baselines.json.  bench.py attaches the entry of a workload (marked "where": "build
container") whenever its live CPU leg is switched off, so no committed bench line goes without a CPU figure
(VERDICT r4 "missing" 5).

    python tools/cpu_baselines.py [workload ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("PBC_CPU_SAMPLE_SCALE", "8")       # 8 vCPU here against 256 on the GPU box: 5-15 s per workload
    import bench
    import bench_group
    import pbc_amd
    names = sys.argv[1:] or sorted(bench.WORKLOADS) + sorted(bench_group.GROUP_WORKLOADS)
    out_path = bench.CPU_BASELINE_FILE
    doc = {"workloads": {}}
    if os.path.exists(out_path):
        doc = json.load(open(out_path))
    model = "?"
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            model = line.split(":", 1)[1].strip()
            break
    doc.update({"cpu_model": model, "cores": os.cpu_count(),
                "commit": subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
                "note": "unmodified reference (oracle/_ref/ref_tool bench / benchg, one forked worker per logical CPU), GMP 6.2.1"})
    for w in names:
        if w in bench.WORKLOADS:
            pname, fixture, k, _, _ = bench.WORKLOADS[w]
            b = bench.cpu_baseline(os.path.join(ROOT, "pbc_amd", "param", pname + ".param"), k, fixture, pp=w.endswith("-pp"))
        else:
            pname, fixture, op, _ = bench_group.GROUP_WORKLOADS[w]
            b = bench_group.cpu_baseline(os.path.join(ROOT, "pbc_amd", "param", pname + ".param"), op)
        doc["workloads"][w] = b
        print(w, json.dumps(b), flush=True)
        json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
