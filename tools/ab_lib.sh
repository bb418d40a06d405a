#!/bin/bash
# same-box A/B of library variants: tools/ab_lib.sh "<lib> ..." "<workload> ..."
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT" || exit 1
for rep in 1 2; do for w in $2; do for l in $1; do
  PBC_HIP_LIB=$l timeout 300 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', '$w', j['value'], j['roofline']['kernel_ms'])
except Exception as e: print('$l $w failed', e)"
done; done; done
