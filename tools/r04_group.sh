#!/bin/sh
# round 4, GPU box: the GPU test-suite, then one bench line per group-operation workload (bench_group.py) under gpurun_out/r04_group/
mkdir -p gpurun_out/r04_group
python -m pytest tests -m gpu -q --maxfail=25 2>&1 | tail -40 > gpurun_out/r04_group/pytest.log
for w in a-g1-mul a-gt-pow a-hash-g1 a-g1-pp a-gt-pp a-bls-verify d-g1-mul d-g2-mul d-gt-pow d-hash-g1 d-g1-pp d-gt-pp f-g1-mul f-g2-mul f-gt-pow f-hash-g1 f-g1-pp f-gt-pp; do
  timeout 300 python bench.py --workload $w --steps 3 > gpurun_out/r04_group/bench_$w.json 2> gpurun_out/r04_group/bench_$w.err || echo "$w failed" >> gpurun_out/r04_group/failed.txt
done
timeout 200 python bench.py --workload a-g1-mul --steps 3 --no-cpu-baseline --param-extra hip_group_slow=1 > gpurun_out/r04_group/bench_a-g1-mul_slow.json 2>> gpurun_out/r04_group/bench_slow.err
timeout 200 python bench.py --workload a-gt-pow --steps 3 --no-cpu-baseline --param-extra hip_group_slow=1 > gpurun_out/r04_group/bench_a-gt-pow_slow.json 2>> gpurun_out/r04_group/bench_slow.err
tail -5 gpurun_out/r04_group/pytest.log
cat gpurun_out/r04_group/failed.txt 2>/dev/null
for f in gpurun_out/r04_group/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print("%-28s %14.1f %s  %.2f ms  frac %s (%s)  cpu %s" % (sys.argv[1].split("bench_")[1][:-5], j["value"], j["unit"], r["kernel_ms"], r["frac"], r["frac_basis"], (j.get("cpu_baseline") or {}).get("value")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
