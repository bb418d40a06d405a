#!/bin/bash
# round 6, fifth GPU call: glue throughput on both exchange formats, bench lines with the new warm-up, the sop-VM probe
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R || exit 1
bash tools/boxinfo.sh 2>&1 | head -3 > $O/boxinfo.txt
export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
for p in a d159 f; do for l in 1 0; do PBC_HIP_GLUE_LIMBS=$l timeout 300 oracle/_ref/glue_test pbc_amd/param/$p.param 1048576 bench 2>&1 | tail -n 1 | sed "s/^/limbs=$l /"; done; done > $O/glue.txt
for t in 4 8 16 32; do PBC_HIP_GLUE_THREADS=$t timeout 300 oracle/_ref/glue_test pbc_amd/param/a.param 1048576 bench 2>&1 | tail -n 1 | sed "s/^/threads=$t /"; done >> $O/glue.txt
unset PBC_HIP_LIB
for w in d f a d-pp a-prod16; do timeout 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-host-path > $O/bench_$w.json 2>> $O/bench.err; done
[ -x tools/exp/sopvm_probe ] && timeout 120 tools/exp/sopvm_probe > $O/sopvm.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "glue" 2>&1 | tail -n 3 > $O/pytest_glue.txt
cat $O/glue.txt $O/sopvm.txt $O/pytest_glue.txt; python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6e/bench_*.json")):
    j=json.loads(open(f).read()); print(f.split("bench_")[1], j["value"], j["roofline"]["kernel_ms"], j["warmup_launches"], j.get("clocks"))
P
