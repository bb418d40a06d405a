#!/bin/bash
# round 6, twenty-first GPU call: type f products on wavefronts: tests, latency, through the hooks
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6u; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_fwave.py -m gpu -q 2>&1 | tail -n 25 > $O/pytest_fwave.txt; cat $O/pytest_fwave.txt
for k in 2 4 16; do DW_PARAM=f timeout 300 python tools/dwave_latency.py prod $k 1 64 256 1024 4096 > $O/lat_f_prod$k.txt 2>&1; cat $O/lat_f_prod$k.txt; done
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 200 oracle/_ref/glue_test pbc_amd/param/f.param 100 latency 2>&1 | tail -n 2 | tee $O/glue.txt
timeout 200 oracle/_ref/glue_test pbc_amd/param/f.param 60 2>&1 | tail -n 2
unset PBC_HIP_LIB
