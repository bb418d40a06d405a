#!/bin/bash
# Round-3 evidence on the GPU box (one gpurun call): bench lines of every workload, the instruction probes,
# rocprofv3 kernel traces (--kernel-trace --stats) and PMC passes (separate --pmc runs) of the BASELINE configs.
# BENCH_WL / PMC_WL restrict the workload lists, SKIP_HEAD=1 skips the headline line and the probes (e.g. after a change to one kernel).
# Everything lands under gpurun_out/ev_r03/; tools/r03_summarise.py turns it into profiles/r03_*.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/ev_r03; mkdir -p $O; cd $R || exit 1
[ -z "$WITH_TESTS" ] || { timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log; }
[ -n "$SKIP_HEAD" ] || timeout 400 python bench.py --steps 5 --warmup 1 > $O/bench_a.json 2> $O/bench_a.err
for w in ${BENCH_WL-d f a-prod16 d-prod16 a-pp d-pp g e a1 f256 d190 d201 d224}; do
  NOCPU="--no-cpu-baseline"; case " ${CPU_WL-d f a-prod16} " in *" $w "*) NOCPU="";; esac     # the reference's CPU rate beside the BASELINE configs only
  timeout 400 python bench.py --workload $w --steps 3 --warmup 1 $NOCPU > $O/bench_$w.json 2> $O/bench_$w.err
done
[ -n "$SKIP_HEAD$SKIP_PROBES" ] || timeout 300 python tools/probe.py > $O/probe.txt 2>&1
[ -n "$SKIP_HEAD$SKIP_PROBES" ] || (hipcc --offload-arch=gfx950 -O2 tools/mac_chain_probe.hip -o /tmp/mac_chain 2>/dev/null && timeout 120 /tmp/mac_chain) > $O/mac_chain.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for w in ${PMC_WL-a d f a-prod16}; do
  B="python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-path"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- $B > $O/kt_$w.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc1_$w -- $B > $O/pmc1_$w.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_IFETCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc2_$w -- $B > $O/pmc2_$w.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc3_$w -- $B > $O/pmc3_$w.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc4_$w -- $B > $O/pmc4_$w.log 2>&1
done
cd $R; find $O -name "*.csv" | wc -l; tail -n 1 $O/bench_a.json | cut -c1-200
