#!/bin/bash
# round 6, twenty-seventh GPU call: the wave kernels of type a1 / generic type a (pairing_aw.cuh AG<N>): tests, through the hooks
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6aa; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 1500 python -m pytest tests/test_gpu_agwave.py -m gpu -q -x 2>&1 | tail -n 40 > $O/pytest_agwave.txt; cat $O/pytest_agwave.txt
timeout 600 python -m pytest tests/test_gpu_wave.py -m gpu -q -x 2>&1 | tail -n 5 | tee $O/pytest_wave.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in a1 a_160_1024 a; do timeout 300 oracle/_ref/glue_test pbc_amd/param/$p.param 20 latency 2>&1 | tail -n 2; done | tee $O/glue.txt
unset PBC_HIP_LIB
