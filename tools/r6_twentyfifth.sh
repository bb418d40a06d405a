#!/bin/bash
# round 6, twenty-fifth GPU call: type g products and pairing_pp_apply on wavefronts: tests, latency, through the hooks
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6y; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 1200 python -m pytest tests/test_gpu_gwave.py -m gpu -q -x 2>&1 | tail -n 40 > $O/pytest_gwave.txt; cat $O/pytest_gwave.txt
DW_PARAM=g149 timeout 300 python tools/dwave_latency.py prod 4 1 16 256 1024 4096 > $O/lat_g_prod4.txt 2>&1; cat $O/lat_g_prod4.txt
DW_PARAM=g149 timeout 300 python tools/dwave_latency.py prod 16 1 64 512 > $O/lat_g_prod16.txt 2>&1; cat $O/lat_g_prod16.txt
DW_PARAM=g149 timeout 300 python tools/dwave_latency.py pp 1 16 256 1024 4096 8192 > $O/lat_g_pp.txt 2>&1; cat $O/lat_g_pp.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 200 oracle/_ref/glue_test pbc_amd/param/g149.param 60 latency 2>&1 | tail -n 2 | tee $O/glue.txt
unset PBC_HIP_LIB
