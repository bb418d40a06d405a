#!/bin/bash
# round 6, first GPU call: box identity, the new tests (soak, aliasing, two threads on a stream), bench lines with the clock sampler
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6a; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh > $O/boxinfo.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_soak.py tests/test_gpu_group2.py tests/test_gpu_wave.py -m gpu -q -x --durations=8 > $O/pytest_new.log 2>&1; tail -n 15 $O/pytest_new.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "prod" > $O/pytest_prod.log 2>&1; tail -n 3 $O/pytest_prod.log
for w in d a f d; do timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-host-path >> $O/bench_$w.json 2>> $O/bench.err; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6a/bench_*.json")):
    for l in open(f):
        j=json.loads(l); print(f.split("bench_")[1], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["peak_measured"], j.get("clocks"))
P
