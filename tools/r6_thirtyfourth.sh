#!/bin/bash
# round 6, thirty-fourth GPU call: the abort of the closing collection's suite (test_gpu_dwave, products around the cut-over): alone, then the whole suite again
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6ah; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_dwave.py -m gpu -q > $O/pytest_dwave.txt 2>&1; tail -n 4 $O/pytest_dwave.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_agwave.py tests/test_gpu_configs.py tests/test_gpu_dwave.py -m gpu -q > $O/pytest_first3.txt 2>&1; tail -n 4 $O/pytest_first3.txt | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -n 4 $O/pytest_all.txt | cut -c1-200
dmesg 2>/dev/null | tail -n 20 > $O/dmesg.txt
