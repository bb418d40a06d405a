#!/bin/bash
# round 6, eighteenth GPU call: soak of the wave kernels (random + crafted units from the compiled reference), all fields
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6r; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_soak.py -m gpu -q -k wave 2>&1 | tail -n 15 > $O/pytest_soak_wave.txt; cat $O/pytest_soak_wave.txt
PBC_SOAK_SEED=777003 PBC_SOAK_LOG2_WAVE=13 timeout 900 python -m pytest tests/test_gpu_soak.py -m gpu -q -k wave 2>&1 | tail -n 5 >> $O/pytest_soak_wave.txt; tail -n 3 $O/pytest_soak_wave.txt
