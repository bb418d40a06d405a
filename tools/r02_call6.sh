#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/r02f"
mkdir -p "$OUT"
cd "$ROOT" || exit 1
run() {  # lib workload
  PBC_HIP_LIB=$1 timeout 300 python bench.py --workload $2 --steps 4 --warmup 1 --no-cpu-baseline --no-host-path 2> "$OUT/bench_$2_$1.err" | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline']['frac_basis'])
except Exception as e: print('$1 $2 failed', e)"
}
run libpbc_hip.so a-prod16
for w in d g f d190 d-prod16; do run libpbc_hip.so $w; run libpbc_hip_dfw3.so $w; done
timeout 600 python -m pytest tests -m gpu -q -k "unmodified or glue or all_zero or two_ranks" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest subset: exit $?"; tail -n 4 "$OUT/pytest_gpu.log"
