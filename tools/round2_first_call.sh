#!/bin/bash
# First GPU call of the next round, in one gpurun (about 8 minutes of box time):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/round2_first_call.sh'
# 1. the GPU parity suite (the x-only / twist tests and the glue's from_hash check have only run on the host
#    mirror so far), 2. parity + A/B timing of the signed-limb type d kernels (PBC_HIP_D_LAZY), 3. the type d
#    bench line with both settings, 4. a kernel trace of the experiment.  Everything lands in gpurun_out/r02/.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/r02"
mkdir -p "$OUT"
cd "$ROOT" || exit 1
timeout 600 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest -m gpu: exit $?" | tee -a "$OUT/pytest_gpu.log"
PBC_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests -m gpu -q -k signed_limb > "$OUT/pytest_experiment.log" 2>&1
echo "experiment test: exit $?" | tee -a "$OUT/pytest_experiment.log"
timeout 300 python tools/d_lazy_ab.py 18 > "$OUT/d_lazy_ab.jsonl" 2>&1
cat "$OUT/d_lazy_ab.jsonl"
for lz in 0 1; do
  PBC_HIP_D_LAZY=$lz timeout 300 python bench.py --workload d --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_d_lazy$lz.json" 2> "$OUT/bench_d_lazy$lz.err"
  tail -n 1 "$OUT/bench_d_lazy$lz.json"
  PBC_HIP_D_LAZY=$lz timeout 300 python bench.py --workload d-pp --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_dpp_lazy$lz.json" 2> "$OUT/bench_dpp_lazy$lz.err"
  tail -n 1 "$OUT/bench_dpp_lazy$lz.json"
done
cd /tmp && export TMPDIR=/tmp
PBC_HIP_D_LAZY=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_d_lazy" -- python "$ROOT/bench.py" --workload d --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/prof_d_lazy.log" 2>&1
echo "done"
