#!/bin/bash
# round 6, thirteenth GPU call: the whole suite with products / pairing_pp_apply of small d159 batches on wavefronts; the hooks' latency
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6m; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
timeout 200 oracle/_ref/glue_test pbc_amd/param/d159.param 200 latency > $O/glue.txt 2>&1; tail -n 3 $O/glue.txt
timeout 200 oracle/_ref/glue_test pbc_amd/param/a.param 200 latency >> $O/glue.txt 2>&1; tail -n 2 $O/glue.txt
unset PBC_HIP_LIB
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 6 > $O/pytest_all.txt; cat $O/pytest_all.txt
