#!/bin/bash
# round 6, seventeenth GPU call: the wave kernels on the seven-word type d fields (eight limbs); latency; through the hooks
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6q; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_dwave.py -m gpu -q 2>&1 | tail -n 15 > $O/pytest_dwave.txt; cat $O/pytest_dwave.txt
for p in d201 d224; do
  DW_PARAM=$p timeout 300 python tools/dwave_latency.py 1 256 1024 2048 4096 8192 > $O/lat_$p.txt 2>&1; cat $O/lat_$p.txt
done
DW_PARAM=d224 timeout 300 python tools/dwave_latency.py prod 4 1 256 > $O/lat_prod4_d224.txt 2>&1; cat $O/lat_prod4_d224.txt
DW_PARAM=d224 timeout 300 python tools/dwave_latency.py pp 1 256 > $O/lat_pp_d224.txt 2>&1; cat $O/lat_pp_d224.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in d201 d224; do timeout 200 oracle/_ref/glue_test pbc_amd/param/$p.param 100 latency 2>&1 | tail -n 2 | tee -a $O/glue.txt; done
unset PBC_HIP_LIB
