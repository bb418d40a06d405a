#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/r02h"
mkdir -p "$OUT"
cd "$ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest -m gpu: exit $?" | tee -a "$OUT/pytest_gpu.log"
tail -n 8 "$OUT/pytest_gpu.log"
PBC_HIP_LIB=$ROOT/pbc_amd/libpbc_hip.so timeout 300 oracle/_ref/glue_test pbc_amd/param/a.param 1048576 bench 2>&1 | tail -2
PBC_HIP_LIB=$ROOT/pbc_amd/libpbc_hip.so timeout 300 oracle/_ref/glue_test pbc_amd/param/d159.param 1048576 bench 2>&1 | tail -2
