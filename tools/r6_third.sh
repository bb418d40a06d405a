#!/bin/bash
# round 6, third GPU call: the limb-image route (library tests, glue on both exchange formats, glue throughput), type f with the
# line product's factor switch, d159 with the clock sampler
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "limb_image or glue" 2>&1 | tail -n 4 > $O/pytest_limbs.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in a d159 f; do for l in 1 0; do PBC_HIP_GLUE_LIMBS=$l timeout 300 oracle/_ref/glue_test pbc_amd/param/$p.param 1048576 bench 2>&1 | tail -n 1 | sed "s/^/limbs=$l /"; done; done > $O/glue.txt
unset PBC_HIP_LIB
for w in f d a; do timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_$w.json 2>> $O/bench.err; done
cat $O/pytest_limbs.txt $O/glue.txt; python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6c/bench_*.json")):
    j=json.loads(open(f).read()); print(f.split("bench_")[1], j["value"], j["roofline"]["kernel_ms"], j.get("clocks"))
P
[ -x tools/exp/sopvm_probe ] && timeout 120 tools/exp/sopvm_probe > $O/sopvm.txt 2>&1; cat $O/sopvm.txt
