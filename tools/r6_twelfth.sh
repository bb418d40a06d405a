#!/bin/bash
# round 6, twelfth GPU call: products (one wavefront per term) and pairing_pp_apply on the d159 wave kernels: tests, latency tables
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6l; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_dwave.py -m gpu -q 2>&1 | tail -n 15 > $O/pytest_dwave.txt; cat $O/pytest_dwave.txt
timeout 300 python tools/dwave_latency.py pp 1 256 1024 2048 4096 5120 8192 > $O/pp_latency.txt 2>&1; cat $O/pp_latency.txt
for k in 2 4 16; do timeout 300 python tools/dwave_latency.py prod $k 1 16 64 256 1024 4096 8192 > $O/prod${k}_latency.txt 2>&1; cat $O/prod${k}_latency.txt; done
timeout 300 python tools/dwave_latency.py prod 64 1 16 256 1024 > $O/prod64_latency.txt 2>&1; cat $O/prod64_latency.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 120 oracle/_ref/glue_test pbc_amd/param/d159.param 120 2>&1 | tail -n 2
unset PBC_HIP_LIB
