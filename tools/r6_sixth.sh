#!/bin/bash
# round 6, sixth GPU call: the one-pairing-per-wavefront d159 kernel (tests, latency, through the hooks), the limb route once more, the whole GPU suite
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6f; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 600 python -m pytest tests/test_gpu_dwave.py -m gpu -q -x 2>&1 | tail -n 8 > $O/pytest_dwave.txt; cat $O/pytest_dwave.txt
timeout 300 python tools/dwave_latency.py 1 16 256 1024 2048 4096 8192 > $O/dwave_latency.txt 2>&1; cat $O/dwave_latency.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 120 oracle/_ref/glue_test pbc_amd/param/d159.param 200 latency 2>&1 | tail -n 1 > $O/glue.txt
for l in 1 0; do PBC_HIP_GLUE_LIMBS=$l timeout 300 oracle/_ref/glue_test pbc_amd/param/a.param 1048576 bench 2>&1 | tail -n 2 | sed "s/^/limbs=$l /"; done >> $O/glue.txt
unset PBC_HIP_LIB; cat $O/glue.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 6 > $O/pytest_all.txt; cat $O/pytest_all.txt
