#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs, kernel trace only) of one bench workload: [PASSES="1 2"] [PBC_HIP_LIB=variant.so] tools/r03_pmc.sh <workload> [tag]
R="${GRAFT_REPO_ROOT:-/root/repo}"; W=${1:-f}; TAG=${2:-$W}
O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-host-path"
[[ " ${PASSES-1 2 3 4} " == *" 1 "* ]] && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
[[ " ${PASSES-1 2 3 4} " == *" 2 "* ]] && timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_IFETCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
[[ " ${PASSES-1 2 3 4} " == *" 3 "* ]] && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1
[[ " ${PASSES-1 2 3 4} " == *" 4 "* ]] && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- $B > $O/p4.log 2>&1
cd $R
python - "$O" <<'PY'
import csv, glob, sys, json, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", ""); 
        if "probe" in k or "init" in k: continue
        out[k.split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}
json.dump(res, open(sys.argv[1] + "/summary.json", "w"), indent=1, sort_keys=True)
for k, d in res.items():
    if d.get("SQ_WAVES", 0) >= 1024 or len(res) == 1:
        print(k); print("  " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(d.items())))
PY
