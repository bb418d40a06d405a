#!/bin/bash
# round 6, twenty-sixth GPU call: the whole -m gpu suite with type g products / pairing_pp_apply on wavefronts and the g149 soak case;
# single calls through the hooks on a1.param / e.param (lane kernels only: what a wave kernel would have to beat)
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6z; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 2400 python -m pytest tests -m gpu -q --maxfail 10 > $O/pytest.log 2>&1; tail -n 15 $O/pytest.log
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in a1 e; do timeout 300 oracle/_ref/glue_test pbc_amd/param/$p.param 6 latency 2>&1 | tail -n 2; done | tee $O/glue_a1_e.txt
unset PBC_HIP_LIB
cp gpurun_out/soak_wave-g149_seed*.json $O/ 2>/dev/null
