#!/bin/bash
# round 4, one GPU box: same-box A/B of (a) the type f prefetch of buffered coefficients (library variants), (b) dynamic
# unit fetch and time-sliced priorities (parameter-text switches), (c) the batch-size sweep.  Lines go to gpurun_out/r04_ab/.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT" || exit 1
O=gpurun_out/r04_ab; mkdir -p $O
run() {   # lib workload param_extra
  PBC_HIP_LIB=$1 timeout 300 python bench.py --workload $2 --steps 4 --warmup 1 --no-cpu-baseline --no-host-path ${3:+--param-extra $3} 2>>$O/err.log | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', '${3:-default}', j['value'], j['roofline']['kernel_ms'])
except Exception as e: print('$1 $2 ${3:-default} failed', e)"
}
{
for rep in 1 2; do
  for l in libpbc_hip.so libpbc_hip_nopf.so; do run $l f; done
  for w in a f a-pp d201; do for x in "" hip_dynamic=1 hip_dynamic=1,hip_no_fair=1 hip_no_fair=1; do run libpbc_hip.so $w $x; done; done
  run libpbc_hip.so d
  for x in "" hip_dynamic=1 hip_dynamic=1,hip_no_fair=1; do run libpbc_hip_res5.so d $x; done
done
for x in "" hip_dynamic=1,hip_no_fair=1; do run libpbc_hip.so a-prod16 $x; run libpbc_hip.so a-g1-mul $x; done
} | tee $O/ab.txt
for w in a f; do python bench.py --workload $w --sweep --steps 3 > $O/sweep_$w.json 2>>$O/err.log; done
python bench.py --workload a --sweep --steps 3 --param-extra hip_dynamic=1,hip_no_fair=1 > $O/sweep_a_dynamic.json 2>>$O/err.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f_ or _f_ or df_ or bn_" 2>&1 | tail -3
