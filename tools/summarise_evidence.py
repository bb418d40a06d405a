"""Turn gpurun_out/ev (written by tools/collect_evidence.sh on the GPU box) into the committed
summaries under profiles/ (named per round)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
more = len(sys.argv) > 2 and sys.argv[2] == "more"      # tools/collect_evidence_more.sh output
ev = "gpurun_out/ev2" if more else "gpurun_out/ev"
os.makedirs("profiles", exist_ok=True)
for w in ("d190", "d201", "d224", "g", "e", "a1", "d-pp", "g-pp") if more else ("a", "d", "f", "a-prod16", "a-pp"):
    src = "%s/bench_%s.json" % (ev, w)
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, "profiles/%s_bench_%s.json" % (tag, w))
    ks = glob.glob("%s/kt_%s/**/*kernel_stats.csv" % (ev, w), recursive=True)
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        with open("profiles/%s_kernel_stats_%s.csv" % (tag, w), "w") as fh:
            fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in rows[:6]:
                fh.write('"%s",%s,%s,%s,%s,%s,%s\n' % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                      r["Percentage"], r["MinNs"], r["MaxNs"]))
if os.path.exists(ev + "/probe.txt"):
    shutil.copy(ev + "/probe.txt", "profiles/%s_probe.txt" % tag)
out = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mem"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (ev, name), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "a_pairing_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"launches": len(v), "avg_per_launch": sum(v) / len(v)}
if out:
    json.dump(out, open("profiles/%s_a_pairing_pmc.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
