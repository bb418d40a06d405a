#!/bin/bash
# round 6, twenty-eighth GPU call: the wave kernels of type a1 / generic type a: tests again, the cut-over sweep
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6ab; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 1500 python -m pytest tests/test_gpu_agwave.py -m gpu -q 2>&1 | tail -n 12 > $O/pytest_agwave.txt; cat $O/pytest_agwave.txt
{ AG_PARAM=a1 timeout 600 python tools/agwave_latency.py 1 64 512 1024 2048 4096 8192 16384 32768
  AG_PARAM=a1 AG_WAVE4_MAX=0 timeout 300 python tools/agwave_latency.py 1 512 1024
  AG_PARAM=a1 AG_WAVE4_MAX=100000 LANE_MAX=0 timeout 300 python tools/agwave_latency.py 2048 4096
  AG_PARAM=a1 timeout 300 python tools/agwave_latency.py prod 4 1 256 2048
  AG_PARAM=a1 timeout 300 python tools/agwave_latency.py pp 1 1024 4096 16384
  AG_PARAM=a_160_1024 timeout 300 python tools/agwave_latency.py 1 1024 4096 16384
  AG_PARAM=a_160_512_mm timeout 300 python tools/agwave_latency.py 1 1024 2048 4096 8192
  AG_PARAM=a_160_256 timeout 300 python tools/agwave_latency.py 1 1024 2048 4096 8192; } 2>&1 | grep -v amdgpu.ids | tee $O/agwave_latency.txt
