#!/bin/bash
# round 6, ninth GPU call: the d159 wave kernel as an interpreter of a host-built schedule (tests, latency, hooks, PMC)
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6i; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 600 python -m pytest tests/test_gpu_dwave.py -m gpu -q -x 2>&1 | tail -n 4 > $O/pytest_dwave.txt; cat $O/pytest_dwave.txt
timeout 300 python tools/dwave_latency.py 1 16 256 1024 2048 3072 4096 > $O/dwave_latency.txt 2>&1; cat $O/dwave_latency.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 120 oracle/_ref/glue_test pbc_amd/param/d159.param 200 latency 2>&1 | tail -n 1 > $O/glue.txt; cat $O/glue.txt
unset PBC_HIP_LIB
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_dw -- python $R/tools/dwave_latency.py 1024 > $O/pmc_dw.log 2>&1
cd $R; python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/r6i/pmc_dw/**/*counter_collection.csv", recursive=True):
    by = {}
    for r in csv.DictReader(open(f)):
        if "dw_pairing" in r["Kernel_Name"]: by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print({k: (len(v), sum(v) / len(v) / 1024) for k, v in by.items()})
P
