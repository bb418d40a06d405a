#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/r02e"
mkdir -p "$OUT"
cd "$ROOT" || exit 1
for w in a-prod16 a; do
  timeout 300 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_${w}.json" 2> "$OUT/bench_${w}.err"
  python - "$OUT/bench_${w}.json" "$w" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], j["roofline"]["frac_basis"], (j.get("host_path") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest -m gpu: exit $?" | tee -a "$OUT/pytest_gpu.log"
tail -n 8 "$OUT/pytest_gpu.log"
echo done
