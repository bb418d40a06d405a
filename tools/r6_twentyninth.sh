#!/bin/bash
# round 6, twenty-ninth GPU call: the wave kernels of type a1 / generic type a with the measured cut-overs: their tests, the suites that use a1 / generic a
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6ac; mkdir -p $O; cd $R || exit 1
timeout 1500 python -m pytest tests/test_gpu_agwave.py -m gpu -q 2>&1 | tail -n 12 > $O/pytest_agwave.txt; cat $O/pytest_agwave.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_wave.py -m gpu -q 2>&1 | tail -n 8 | tee $O/pytest_rest.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in a1 a_160_1024 a_160_512_mm; do timeout 300 oracle/_ref/glue_test pbc_amd/param/$p.param 20 latency 2>&1 | tail -n 2; done | tee $O/glue.txt
unset PBC_HIP_LIB
