#!/bin/bash
# round 6, thirty-third GPU call: eight wavefronts per unit on a1 / generic type a (three rounds a doubling step): tests, latency
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6ag; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 1500 python -m pytest tests/test_gpu_agwave.py -m gpu -q 2>&1 | tail -n 12 | tee $O/pytest_agwave.txt
{ AG_PARAM=a1 LANE_MAX=1 timeout 300 python tools/agwave_latency.py 1 16 64 128
  AG_PARAM=a1 AG_WAVE8_MAX=0 LANE_MAX=0 timeout 300 python tools/agwave_latency.py 1 16 64 128
  AG_PARAM=a1 AG_WAVE8_MAX=1000 LANE_MAX=0 timeout 300 python tools/agwave_latency.py 256 512
  AG_PARAM=a1 LANE_MAX=0 timeout 300 python tools/agwave_latency.py prod 4 1 16
  AG_PARAM=a_160_1024 LANE_MAX=0 timeout 300 python tools/agwave_latency.py 1 64 128
  AG_PARAM=a_160_1024 AG_WAVE8_MAX=0 LANE_MAX=0 timeout 300 python tools/agwave_latency.py 1 64 128
  AG_PARAM=a_160_256 LANE_MAX=0 timeout 300 python tools/agwave_latency.py 1 64 128
  AG_PARAM=a_160_256 AG_WAVE8_MAX=0 LANE_MAX=0 timeout 300 python tools/agwave_latency.py 1 64 128; } 2>&1 | grep -v amdgpu.ids | tee $O/agwave8_latency.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 300 oracle/_ref/glue_test pbc_amd/param/a1.param 30 latency 2>&1 | tail -n 2 | tee $O/glue.txt
unset PBC_HIP_LIB
