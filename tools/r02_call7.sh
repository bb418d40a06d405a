#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/r02g"
mkdir -p "$OUT"
cd "$ROOT" || exit 1
run() {  # lib workload
  PBC_HIP_LIB=$1 timeout 300 python bench.py --workload $2 --steps 4 --warmup 1 --no-cpu-baseline --no-host-path 2> "$OUT/bench_$2_$1.err" | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline']['frac_basis'])
except Exception as e: print('$1 $2 failed', e)"
}
for w in f f256; do run libpbc_hip.so $w; done
timeout 900 python -m pytest tests -m gpu -q -x -k "f_ or _f or bn_ or type_f or df_ or bilinearity or group_ops or twist or bls" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest subset: exit $?"; tail -n 4 "$OUT/pytest_gpu.log"
