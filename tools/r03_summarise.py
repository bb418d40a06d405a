"""gpurun_out/ev_r03 (tools/r03_evidence.sh, run on the GPU box) -> committed summaries under profiles/r03_*:
bench lines, kernel-trace stats of the dominant kernel, PMC counters per launch (averaged over the launches of the
workload's main kernel; FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them)."""
import collections
import csv
import glob
import json
import os
import shutil

EV = "gpurun_out/ev_r03"
os.makedirs("profiles", exist_ok=True)
for f in glob.glob(EV + "/bench_*.json"):
    if os.path.getsize(f):
        shutil.copy(f, "profiles/r03_" + os.path.basename(f))
if os.path.exists(EV + "/probe.txt"):
    shutil.copy(EV + "/probe.txt", "profiles/r03_probe.txt")
if os.path.exists(EV + "/mac_chain.txt"):
    shutil.copy(EV + "/mac_chain.txt", "profiles/r03_mac_chain_probe.txt")
MAIN = {"a": "al_pairing_kernel", "d": "d_prod_pairing_kernel", "f": "f_prod_pairing_kernel", "a-prod16": "al_miller_kernel"}
for w, kern in MAIN.items():
    ks = glob.glob("%s/kt_%s/**/*kernel_stats.csv" % (EV, w), recursive=True)
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        with open("profiles/r03_kernel_stats_%s.csv" % w, "w") as fh:
            fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in rows[:6]:
                fh.write('"%s",%s,%s,%s,%s,%s,%s\n' % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                      r["Percentage"], r["MinNs"], r["MaxNs"]))
    out = {}
    for f in glob.glob("%s/pmc?_%s/**/*counter_collection.csv" % (EV, w), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v = [x for x in v if x >= 0.5 * max(v)]       # the timed full-size launches, not bench.py's small gate launches
            out[k] = {"launches": len(v), "avg_per_launch": sum(v) / len(v)}
    if out:
        json.dump(out, open("profiles/r03_pmc_%s.json" % w, "w"), indent=1, sort_keys=True)
        print(w, {k: "%.4g" % v["avg_per_launch"] for k, v in sorted(out.items())})
