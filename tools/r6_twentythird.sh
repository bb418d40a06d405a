#!/bin/bash
# round 6, twenty-third GPU call: single calls through the hooks for the families without wave kernels (g149, e, a1)
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6w; mkdir -p $O; cd $R || exit 1
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in g149 e a1; do timeout 300 oracle/_ref/glue_test pbc_amd/param/$p.param 20 latency 2>&1 | tail -n 2 | tee -a $O/glue.txt; done
