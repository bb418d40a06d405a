#!/bin/bash
# one bench line per pairing workload (value, ms, executed frac): a quick regression screen against the previous round
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT" || exit 1
for w in ${1:-a d f a-prod16 d-prod16 a-pp d-pp g g-pp e a1 a1-pp f256 d190 d201 d224}; do
  timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%-9s %12.1f %9.3f ms  frac %.4f (%s)  executed %s' % ('$w', j['value'], r['kernel_ms'], r['frac'], r['frac_basis'], r['executed'] and r['executed']['frac']))
except Exception as e: print('$w failed', e)"
done
