#!/bin/bash
# round 6, nineteenth GPU call: one large soak (2^18 random pairs per BASELINE configuration, 2^14 16-term products, 2^14 units on the wave kernels), fresh seed
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6s; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
PBC_SOAK_SEED=777004 PBC_SOAK_LOG2=18 PBC_SOAK_LOG2_PROD=14 PBC_SOAK_LOG2_WAVE=14 timeout 1700 python -m pytest tests/test_gpu_soak.py -m gpu -q 2>&1 | tail -n 8 > $O/pytest_soak_big.txt; cat $O/pytest_soak_big.txt
