#!/bin/bash
# round 6, thirty-first GPU call: type e on the wave routines with wave 0 alone on the final power and the measured cut-overs; the a1 / g149 soak cases
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6ae; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 1500 python -m pytest tests/test_gpu_agwave.py -m gpu -q 2>&1 | tail -n 12 | tee $O/pytest_agwave.txt
timeout 1500 python -m pytest tests/test_gpu_soak.py -m gpu -q -k "wave_kernels and (a1 or g149)" 2>&1 | tail -n 12 | tee $O/pytest_soak.txt
{ AG_PARAM=e AG_WAVE4_MAX=100000 timeout 600 python tools/agwave_latency.py 1 64 512 1024 2048 4096
  AG_PARAM=e AG_WAVE4_MAX=0 LANE_MAX=0 timeout 300 python tools/agwave_latency.py 1 512 1024 2048 4096 8192 16384
  AG_PARAM=e timeout 300 python tools/agwave_latency.py prod 4 1 256 2048
  AG_PARAM=e_160_400 timeout 300 python tools/agwave_latency.py 1 1024 4096 8192; } 2>&1 | grep -v amdgpu.ids | tee $O/ewave_latency.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in e; do timeout 300 oracle/_ref/glue_test pbc_amd/param/$p.param 20 latency 2>&1 | tail -n 2; done | tee $O/glue.txt
unset PBC_HIP_LIB
cp gpurun_out/soak_wave-a1_seed*.json gpurun_out/soak_wave-g149_seed*.json $O/ 2>/dev/null
