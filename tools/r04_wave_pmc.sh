#!/bin/bash
# PMC pass + kernel trace of the small-batch kernels (tools/r04_wave.py 256 1024): instructions per pairing of aw_pairing_kernel<16, 4> / <16, 1>
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/wave_pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
B="python $R/tools/r04_wave.py 256 1024"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
cd $R
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for f in glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print("%-70s calls %s avg %.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "aw_pairing" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(agg.items()):
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print(k, "grid", g, "waves %.0f" % w, "  ".join("%s/wave=%.4g" % (c, sum(v) / len(v) / w) for c, v in sorted(d.items()) if c != "SQ_WAVES"))
PY
