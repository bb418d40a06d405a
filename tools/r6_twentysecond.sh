#!/bin/bash
# round 6, twenty-second GPU call: type d products with two Miller steps per visit of a term: tests, A/B (libpbc_hip_nopair.so = one step per visit), PMC traffic
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6v; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dwave.py -m gpu -q -x -k "prod or product" 2>&1 | tail -n 6 > $O/pytest_prod.txt; cat $O/pytest_prod.txt
for rep in 1 2; do for lib in libpbc_hip.so libpbc_hip_nopair.so; do
  [ -f pbc_amd/$lib ] || continue
  PBC_HIP_LIB=$lib timeout 300 python bench.py --workload d-prod16 --steps 6 --warmup 2 --no-cpu-baseline 2> $O/err.txt | tail -n 1 > $O/bench_$lib.$rep.json
  python -c "import json; d=json.loads(open('$O/bench_$lib.$rep.json').read()); print('$lib', d['value'], d['unit'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))" || tail -n 5 $O/err.txt
done; done
for w in d201-prod d-prod4; do :; done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload d-prod16 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path"
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_dprod -- $B > $O/pmc_dprod.log 2>&1
python - <<'P'
import csv,glob,collections
for f in glob.glob('/root/repo/gpurun_out/r6v/pmc_dprod/*/*counter_collection.csv')+glob.glob('/root/repo/gpurun_out/r6v/pmc_dprod/*counter_collection.csv'):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'd_prod' in k: print(k, {c:(len(x), sum(x)/len(x)) for c,x in v.items()})
P
