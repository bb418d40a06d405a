#!/bin/bash
# same-box A/B of parameter-text switches: tools/r04_ab2.sh "<workload> ..." "<extra> ..." (extra "-" = default)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT" || exit 1
for rep in 1 2; do for w in $1; do for x in $2; do
  [ "$x" = "-" ] && X="" || X="--param-extra $x"
  timeout 300 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-host-path $X 2>/dev/null | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$x', j['value'], j['roofline']['kernel_ms'])
except Exception as e: print('$w $x failed', e)"
done; done; done
