#!/bin/bash
# round 6, fifteenth GPU call: joint limb-form ladders for pow2 / pow3 on G1 of d159 / f: tests, A/B with the composition
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6o; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_group2.py -m gpu -q -x 2>&1 | tail -n 15 > $O/pytest_group2.txt; cat $O/pytest_group2.txt
for w in a-g1-pow2 d-g1-pow2 f-g1-pow2; do for x in "" "hip_multi_compose=1"; do
  timeout 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline ${x:+--param-extra $x} 2> $O/err.txt | tail -n 1 > $O/bench_${w}_${x:-joint}.json
  python -c "import json,sys; d=json.loads(open('$O/bench_${w}_${x:-joint}.json').read()); print('$w ${x:-joint}', d['value'], d['unit'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))" || tail -n 5 $O/err.txt
done; done
