#!/bin/bash
# round 6, twentieth GPU call: the type f wave kernel: tests, latency, through the hooks
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6t; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_fwave.py -m gpu -q 2>&1 | tail -n 25 > $O/pytest_fwave.txt; cat $O/pytest_fwave.txt
DW_PARAM=f timeout 300 python tools/dwave_latency.py 1 16 256 1024 2048 4096 8192 > $O/lat_f.txt 2>&1; cat $O/lat_f.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 200 oracle/_ref/glue_test pbc_amd/param/f.param 100 latency 2>&1 | tail -n 2 | tee $O/glue.txt
unset PBC_HIP_LIB
