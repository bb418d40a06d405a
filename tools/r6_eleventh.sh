#!/bin/bash
# round 6, eleventh GPU call: the d159 wave kernel with the schedule in a buffer the object keeps (tests, latency, through the hooks), limb calls, the whole suite
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6k; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 600 python -m pytest tests/test_gpu_dwave.py -m gpu -q -x 2>&1 | tail -n 4 > $O/pytest_dwave.txt; cat $O/pytest_dwave.txt
timeout 300 python tools/dwave_latency.py 1 16 256 1024 2048 4096 5120 6144 8192 > $O/dwave_latency.txt 2>&1; cat $O/dwave_latency.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 120 oracle/_ref/glue_test pbc_amd/param/d159.param 200 latency > $O/glue.txt 2>&1; tail -n 3 $O/glue.txt
timeout 120 oracle/_ref/glue_test pbc_amd/param/d159.param 120 2>&1 | tail -n 2
unset PBC_HIP_LIB
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 6 > $O/pytest_all.txt; cat $O/pytest_all.txt
