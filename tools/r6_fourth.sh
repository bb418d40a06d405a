#!/bin/bash
# round 6, fourth GPU call: the limb-image route under the microscope; d159 with and without the profiler on ONE box
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6d; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 600 python tools/exp/limb_debug.py > $O/limb_debug.txt 2>&1; cat $O/limb_debug.txt | tail -n 14
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "limb_image" 2>&1 | tail -n 6 > $O/pytest_limbs.txt; cat $O/pytest_limbs.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in a d159; do PBC_HIP_VERBOSE=1 timeout 300 oracle/_ref/glue_test pbc_amd/param/$p.param 120 2>&1 | tail -n 4; done > $O/glue120.txt; cat $O/glue120.txt
unset PBC_HIP_LIB
# the same command on the same box: plain, under --kernel-trace --stats, under a PMC pass (VERDICT r5 "weak" 2: 17.1 ms in the profile, 14.1 in the bench line)
B="python $R/bench.py --workload d --steps 5 --warmup 2 --no-cpu-baseline --no-host-path"
for i in 1 2; do $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('plain', j['roofline']['kernel_ms'], j['clocks'])"; done > $O/d_plain.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_d -- $B > $O/kt_d.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_d -- $B > $O/pmc_d.log 2>&1
cd $R
python - <<'P'
import csv, glob, json
O = "gpurun_out/r6d"
print(open(O + "/d_plain.txt").read())
for f in glob.glob(O + "/kt_d/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]: print("kernel-trace", r["Name"][:50], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
for f in glob.glob(O + "/kt_d.log"): print([l for l in open(f) if l.startswith("{")][-1][:0])
for f in glob.glob(O + "/pmc_d/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "d_prod_pairing" in r["Kernel_Name"]]
    by = {}
    for r in rows: by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print("pmc", {k: (len(v), max(v)) for k, v in by.items()})
    ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Counter_Name"] == "GRBM_GUI_ACTIVE"] if rows and "Start_Timestamp" in rows[0] else []
    print("pmc durations ms", [round((b - a) / 1e6, 3) for a, b in ts][-5:])
for log in ("kt_d.log", "pmc_d.log"):
    for l in open(O + "/" + log):
        if l.startswith("{"):
            j = json.loads(l); print(log, "bench line under the profiler:", j["roofline"]["kernel_ms"], j["clocks"])
P
