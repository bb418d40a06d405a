#!/bin/bash
# A/B of an environment switch on ONE box: tools/r03_ab_env.sh "VAR=0 VAR=1" "<workload ...>"  (two repetitions, interleaved)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT" || exit 1
for rep in 1 2; do
for w in $2; do for e in $1; do
  env $e timeout 300 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', '$w', j['value'], j['roofline']['kernel_ms'])
except Exception as e: print('$e $w failed', e)"
done; done; done
