#!/bin/bash
# GPU call 2 of round 2: parity suite with the new input classes, the bench line in its new shape, the instruction probes,
# and A/B of register budgets (waves per SIMD) / inlined small-field products for the type d / f / g kernels.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/r02b"
mkdir -p "$OUT"
cd "$ROOT" || exit 1
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1
echo "pytest -m gpu: exit $?" | tee -a "$OUT/pytest_gpu.log"
tail -n 5 "$OUT/pytest_gpu.log"
timeout 300 python bench.py --steps 5 --warmup 1 > "$OUT/bench_a.json" 2> "$OUT/bench_a.err"; tail -n 1 "$OUT/bench_a.json"
timeout 200 python tools/probe.py > "$OUT/probe.txt" 2>&1; head -n 16 "$OUT/probe.txt"
for v in "" _dfw3 _dfw4 _small; do
  for w in d f g d190; do
    PBC_HIP_LIB=libpbc_hip$v.so timeout 200 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-host-path > "$OUT/bench_${w}${v}.json" 2> "$OUT/bench_${w}${v}.err"
    python - "$OUT/bench_${w}${v}.json" "$w$v" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  done
done
echo done
