#!/bin/bash
# round 6, thirty-fifth GPU call: hunting the one abort of a closing collection (test_gpu_dwave, products around the cut-over): the file in a loop, then the
# suite's first three files in a loop, runtime errors logged (AMD_LOG_LEVEL=1); stops at the first failure and keeps its log
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6ai; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
export AMD_LOG_LEVEL=1
fail=0
for i in $(seq 1 30); do timeout 300 python -m pytest tests/test_gpu_dwave.py -m gpu -q -x > $O/loop_dwave.txt 2>&1 || { fail=1; cp $O/loop_dwave.txt $O/FAILED_dwave_$i.txt; break; }; done
echo "dwave loop: $i iterations, fail=$fail" | tee $O/summary.txt
if [ $fail = 0 ]; then for i in $(seq 1 8); do timeout 600 python -m pytest tests/test_gpu_agwave.py tests/test_gpu_configs.py tests/test_gpu_dwave.py -m gpu -q -x > $O/loop_first3.txt 2>&1 || { fail=1; cp $O/loop_first3.txt $O/FAILED_first3_$i.txt; break; }; done; echo "first3 loop: $i iterations, fail=$fail" | tee -a $O/summary.txt; fi
ls $O | grep FAILED | while read f; do grep -v "^  File" $O/$f | tail -n 30 | cut -c1-300; done
