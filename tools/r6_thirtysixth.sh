#!/bin/bash
# round 6, thirty-sixth GPU call: the intermittent abort (suite's first three files, test 150): the same loop with the eight-wavefront route off by default
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6aj; mkdir -p $O; cd $R || exit 1
export LIBC_FATAL_STDERR_=1 AMD_LOG_LEVEL=1
fail=0
for i in $(seq 1 12); do timeout 600 python -X faulthandler -m pytest tests/test_gpu_agwave.py tests/test_gpu_configs.py tests/test_gpu_dwave.py -m gpu -q -x -k "not eight_wavefronts" > $O/loop_first3.txt 2>&1 || { fail=1; cp $O/loop_first3.txt $O/FAILED_first3_$i.txt; break; }; done; echo "first3 loop (hip_wave8_max 0): $i iterations, fail=$fail" | tee $O/summary.txt
ls $O | grep FAILED | while read f; do grep -v "site-packages\|dist-packages" $O/$f | tail -n 40 | cut -c1-300; done
