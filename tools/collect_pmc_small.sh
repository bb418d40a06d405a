#!/bin/bash
# PMC pass (VALU issue counters) for the small-field kernels: types d, f, g.  Output: gpurun_out/ev3
R=$PWD; mkdir -p gpurun_out/ev3; O=$R/gpurun_out/ev3
cd /tmp && export TMPDIR=/tmp
for w in ${@:-d f g}; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_$w -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$w.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $O/pmc2_$w -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc2_$w.log 2>&1
done
cd $R; find gpurun_out/ev3 -name "*counter_collection.csv" | wc -l
