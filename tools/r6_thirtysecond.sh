#!/bin/bash
# round 6, thirty-second GPU call: soak of every route against the compiled reference with two fresh seeds (the second with 2^14 units on the wave kernels)
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6af; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
PBC_SOAK_SEED=777005 timeout 1200 python -m pytest tests/test_gpu_soak.py -m gpu -q 2>&1 | tail -n 6 | tee $O/pytest_soak_a.txt
PBC_SOAK_SEED=777006 PBC_SOAK_LOG2_WAVE=14 timeout 1700 python -m pytest tests/test_gpu_soak.py -m gpu -q -k wave 2>&1 | tail -n 6 | tee $O/pytest_soak_b.txt
cp gpurun_out/soak_*seed777005.json gpurun_out/soak_*seed777006.json $O/ 2>/dev/null; ls $O | wc -l
