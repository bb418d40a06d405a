#!/bin/bash
# round 6, sixteenth GPU call: the wave kernels on the six-word type d fields; latency there; f.param / d190 single calls through the hooks
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6p; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_dwave.py -m gpu -q -x 2>&1 | tail -n 15 > $O/pytest_dwave.txt; cat $O/pytest_dwave.txt
for p in d278027-190-181 d277699-175-167; do
  DW_PARAM=$p timeout 300 python tools/dwave_latency.py 1 256 1024 2048 4096 8192 > $O/lat_$p.txt 2>&1; cat $O/lat_$p.txt
done
DW_PARAM=d278027-190-181 timeout 300 python tools/dwave_latency.py prod 4 1 256 > $O/lat_prod4_d190.txt 2>&1; cat $O/lat_prod4_d190.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in f d278027-190-181 d201; do timeout 200 oracle/_ref/glue_test pbc_amd/param/$p.param 100 latency 2>&1 | tail -n 2 | tee -a $O/glue.txt; done
unset PBC_HIP_LIB
