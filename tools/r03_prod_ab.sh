#!/bin/bash
# Same-box A/B of the two type a product kernels (one term per lane vs one product per lane with the shared squaring)
# plus the GPU tests of the product paths.  Lands under gpurun_out/prod_ab/.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/prod_ab; mkdir -p $O; cd $R || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prod or kat" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
for v in term shared; do
  X=""; [ $v = shared ] && X="--param-extra hip_prod_shared=1"
  timeout 300 python bench.py --workload a-prod16 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path $X > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], j['value'], j['ms_per_step'], j['roofline']['frac'])
except Exception as e: print(sys.argv[1], 'failed', e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
