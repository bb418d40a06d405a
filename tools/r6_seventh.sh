#!/bin/bash
# round 6, seventh GPU call: the d159 wave kernel with row prefetch (tests, latency, phase what-ifs), the limb entry points in the glue's call shape
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6g; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 600 python -m pytest tests/test_gpu_dwave.py -m gpu -q -x 2>&1 | tail -n 4 > $O/pytest_dwave.txt; cat $O/pytest_dwave.txt
timeout 300 python tools/dwave_latency.py 1 16 256 1024 2048 3072 4096 > $O/dwave_latency.txt 2>&1; cat $O/dwave_latency.txt
for v in dw1 dw2 dw3; do echo "what-if $v (1: no Miller loop, 2: no final exponentiation, 3: neither)"; PBC_HIP_LIB=variants/lib$v.so timeout 120 python tools/dwave_latency.py 1 1024 2>&1 | grep "n ="; done > $O/dwave_whatif.txt; cat $O/dwave_whatif.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 120 oracle/_ref/glue_test pbc_amd/param/d159.param 200 latency 2>&1 | tail -n 1 > $O/glue.txt; cat $O/glue.txt
unset PBC_HIP_LIB
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "limb_image_calls_in_a_row" 2>&1 | tail -n 12 > $O/pytest_limb_threads.txt; cat $O/pytest_limb_threads.txt
