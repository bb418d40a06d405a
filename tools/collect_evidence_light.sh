#!/bin/bash
# Reduced version of collect_evidence.sh: benches + rocprofv3 kernel traces of the headline set
# (no probes, no PMC passes).  Output under gpurun_out/ev; tools/summarise_evidence.py <tag>.
R=$PWD; mkdir -p gpurun_out/ev; O=$R/gpurun_out/ev
timeout 400 python bench.py --steps 5 --warmup 1 --host-path > $O/bench_a.json 2> $O/bench_a.err
for w in d f a-prod16 a-pp; do timeout 400 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
for w in a d f a-prod16 a-pp; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- $B --workload $w > $O/kt_$w.log 2>&1; done
cd $R; for w in a d f a-prod16 a-pp; do tail -1 $O/bench_$w.json | cut -c1-170; done
