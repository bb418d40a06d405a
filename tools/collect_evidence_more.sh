#!/bin/bash
# Runs on the GPU box (gpurun): benches + rocprofv3 kernel traces for the curve families added after
# the headline set (other type d widths, types g, a1, e).  Output under gpurun_out/ev2;
# tools/summarise_evidence.py <tag> more turns it into profiles/.
R=$PWD; mkdir -p gpurun_out/ev2; O=$R/gpurun_out/ev2
W="${@:-d190 d201 d224 g e a1}"
for w in $W; do timeout 500 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for w in $W; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- $B --workload $w > $O/kt_$w.log 2>&1; done
cd $R; for w in $W; do tail -1 $O/bench_$w.json | cut -c1-200; done
