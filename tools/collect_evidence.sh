#!/bin/bash
# Runs on the GPU box (gpurun): benches, probes, rocprofv3 kernel trace + PMC passes.
# Everything lands under gpurun_out/; tools/summarise_evidence.py turns it into profiles/.
R=$PWD; mkdir -p gpurun_out/ev; O=$R/gpurun_out/ev
timeout 400 python bench.py --steps 5 --warmup 1 --host-path > $O/bench_a.json 2> $O/bench_a.err
for w in d f a-prod16 a-pp; do timeout 400 python bench.py --workload $w --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 400 python tools/probe.py > $O/probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_a -- $B > $O/kt_a.log 2>&1
for w in d f a-prod16 a-pp; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- $B --workload $w > $O/kt_$w.log 2>&1; done
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mem -- $B > $O/pmc_mem.log 2>&1
cd $R; find gpurun_out/ev -name "*.csv" | wc -l; tail -1 $O/bench_a.json | cut -c1-160
