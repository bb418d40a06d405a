#!/usr/bin/env python3
"""Markdown table of the committed bench lines of a round (profiles/<round>_bench_*.json) beside the previous round's:
value, ms per launch, roofline fraction (basis), executed / algorithmic fractions, HBM traffic ratio, CPU baseline.
    python tools/bench_table.py [r05] [r04]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
prev = sys.argv[2] if len(sys.argv) > 2 else "r04"


def load(r, w):
    p = os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (r, w))
    if not os.path.exists(p):
        return None
    try:
        return json.loads(open(p).readline())
    except Exception:  # noqa: BLE001
        return None


def fmt(v):
    return "%.3g M" % (v / 1e6) if v >= 1e5 else "%.3g k" % (v / 1e3)


names = sorted(os.path.basename(p)[len(rnd) + 7:-5] for p in glob.glob(os.path.join(ROOT, "profiles", rnd + "_bench_*.json")))
order = ["a", "d", "f", "a-prod16", "d-prod16", "a-pp", "d-pp", "g", "g-pp", "d190", "d201", "d224", "e", "a1", "a1-pp", "f256"]
names = [n for n in order if n in names] + [n for n in names if n not in order]
print("| workload | %s | %s | ms / launch | `frac` (basis) | executed / algorithmic | traffic vs records | CPU reference (cores; where) |" % (prev, rnd))
print("|---|---|---|---|---|---|---|---|")
for w in names:
    j, o = load(rnd, w), load(prev, w)
    if not j:
        continue
    r = j["roofline"]
    ex = r.get("executed") or {}
    al = r.get("algorithmic") or {}
    tr = r.get("traffic") or {}
    cb = j.get("cpu_baseline") or {}
    where = cb.get("where", "GPU box") if cb else ""
    print("| `%s` | %s | **%s** %s | %.2f | %s (%s) | %s / %s | %s | %s |" % (
        w, fmt(o["value"]) if o else "--", fmt(j["value"]), j["unit"].split(" per ")[0].split("/")[0] + "/s", r["kernel_ms"],
        r.get("frac"), r.get("frac_basis", ""), ex.get("frac", "--"), al.get("frac", "--"),
        ("%.1fx" % tr["ratio_vs_algorithmic"]) if tr.get("ratio_vs_algorithmic") else "--",
        ("%s %s (%d; %s)" % (fmt(cb["value"]), cb["unit"].split("/")[0] + "/s", cb["cores"], "container" if "container" in where else "GPU box")) if cb else "--"))
