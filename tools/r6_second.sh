#!/bin/bash
# round 6, second GPU call: same-box A/B of the sum-of-products chain and the line product's factor switch (types f, d), the
# clock sampler on the right card, two more soak seeds
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6b; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -4 > $O/boxinfo.txt
bash tools/ab_lib.sh "libpbc_hip.so variants/libfsd2.so variants/libfls1.so variants/libfboth.so" "f" > $O/ab_f.txt 2>&1
bash tools/ab_lib.sh "libpbc_hip.so variants/libdsd2.so" "d d-prod16 d-pp g" > $O/ab_d.txt 2>&1
for l in variants/libfboth.so variants/libdsd2.so; do PBC_HIP_LIB=$l timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "not type_a" 2>&1 | tail -n 2; done > $O/variant_tests.txt
for s in 1 2; do PBC_SOAK_SEED=$((777000 + s)) timeout 600 python -m pytest tests/test_gpu_soak.py -m gpu -q -x 2>&1 | tail -n 1; done > $O/soak_more.txt
timeout 300 python bench.py --workload d --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_d.json 2>> $O/bench.err
cat $O/ab_f.txt $O/ab_d.txt $O/variant_tests.txt $O/soak_more.txt; python -c "
import json; j=json.loads(open('$O/bench_d.json').read()); print(j['value'], j['clocks'])"
