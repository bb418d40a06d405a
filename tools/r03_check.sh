#!/bin/bash
# One gpurun call: the whole GPU suite, then the bench lines named in BENCH_WL (default: a a-prod16; HOSTPATH=1: with the host-buffer path).  Lands under gpurun_out/check/.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/check; mkdir -p $O; cd $R || exit 1
[ -n "$SKIP_TESTS" ] || { timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log; }
HP="--no-host-path"; [ -n "$HOSTPATH" ] && HP=""
for w in ${BENCH_WL-a a-prod16}; do
  timeout 300 python bench.py --workload $w --steps ${STEPS-3} --warmup 1 --no-cpu-baseline $HP > $O/bench_$w.json 2> $O/bench_$w.err
  python - $O/bench_$w.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], j['value'], j['ms_per_step'], j['roofline']['frac'])
except Exception as e: print(sys.argv[1], 'failed', e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
