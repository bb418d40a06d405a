#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for v in unset 0 1; do
  for w in d g; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    timeout 200 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HIP_FORCE_DEV_KERNARG=$v', '$w', j['value'], j['roofline']['kernel_ms'])"
  done
done
