#!/bin/bash
# Round-3 opening call on the GPU box: the GPU parity suite, the headline line and the BASELINE configs, and a
# rocprofv3 kernel trace of the headline command.  Lands under gpurun_out/r03_first/.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r03_first; mkdir -p $O; cd $R || exit 1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout 400 python bench.py --steps 10 --warmup 2 > $O/bench_a.json 2> $O/bench_a.err
for w in ${BENCH_WL-d f a-prod16 d-prod16 g f256}; do
  timeout 400 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_a -- $B > $O/kt_a.log 2>&1
cd $R
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
