#!/bin/bash
# Host-buffer path (pinned host -> host, PCIe included) against the library's switches for it:
#   tools/r03_hostchunk.sh "<workloads>" "<param-extra> ..."     e.g.  "a f" "default hip_host_chunk=262144 hip_zero_copy=1"
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT" || exit 1
for w in ${1-a}; do for x in ${2-default}; do
  X=""; [ "$x" != default ] && X="--param-extra $x"
  timeout 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline $X 2>/dev/null | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $x kernel', j['value'], 'host', j['host_path']['value'], j['host_path']['ms'])
except Exception as e: print('$w $x failed', e)"
done; done
