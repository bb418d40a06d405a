#!/bin/bash
# Evidence, the part that runs ON THE GPU BOX (one gpurun call, started by tools/collect.sh -- which refuses a
# dirty tree and stamps the commit into .evidence_head, so every file written here comes from one commit): the -m gpu
# suite, a bench line per workload, rocprofv3 kernel traces (--kernel-trace --stats) and PMC passes (separate --pmc
# runs, kernel trace only) of the BASELINE configs and the group-operation headliners.
# BENCH_WL / GROUP_WL / PMC_WL restrict the lists; WITH_TESTS=1 runs the suite first.  Output: gpurun_out/ev_$ROUND/.
R="${GRAFT_REPO_ROOT:-/root/repo}"; ROUND=${ROUND:-r06}; O=$R/gpurun_out/ev_$ROUND; mkdir -p $O; cd $R || exit 1
cp .evidence_head $O/HEAD 2>/dev/null || { echo "no .evidence_head: start this through tools/collect.sh"; exit 1; }
[ -z "$WITH_TESTS" ] || { timeout 1500 python -m pytest tests -m gpu -q --maxfail 10 > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log; }
for w in ${BENCH_WL-a d f a-prod16 d-prod16 a-pp d-pp g g-pp e a1 a1-pp f256 d190 d201 d224}; do
  NOCPU="--no-cpu-baseline"; case " ${CPU_WL-a d f a-prod16} " in *" $w "*) NOCPU="";; esac     # live CPU leg beside the BASELINE configs; the others carry the build container's figure (profiles/r05_cpu_baselines.json)
  timeout 400 python bench.py --workload $w --steps 6 --warmup 2 $NOCPU > $O/bench_$w.json 2> $O/bench_$w.err
done
for w in ${GROUP_WL-a-g1-mul a-gt-pow a-hash-g1 a-g1-pp a-gt-pp a-bls-verify a-compress a-decompress a-g1-add a-zr-inv a-g1-pow2 a-gt-pow2 d-g1-mul d-g2-mul d-gt-pow d-hash-g1 d-g1-pp d-gt-pp d-compress d-decompress d-g1-add d-zr-inv d-g1-pow2 d-gt-pow2 f-g1-mul f-g2-mul f-gt-pow f-hash-g1 f-g1-pp f-gt-pp f-compress f-decompress f-g1-add f-zr-inv f-g1-pow2 f-gt-pow2}; do
  NOCPU="--no-cpu-baseline"; case " ${GROUP_CPU_WL-a-g1-mul a-bls-verify} " in *" $w "*) NOCPU="";; esac
  timeout 300 python bench.py --workload $w --steps 6 --warmup 2 $NOCPU > $O/bench_$w.json 2> $O/bench_$w.err
done
[ -n "$SKIP_SWEEP" ] || for w in a f; do timeout 300 python bench.py --workload $w --sweep > $O/sweep_$w.json 2> $O/sweep_$w.err; done
[ -n "$SKIP_SMALL" ] || { timeout 300 python tools/wave_latency.py 1 256 512 1024 2048 4096 5120 > $O/wave_latency.txt 2>&1
  timeout 200 python tools/tail_latency.py > $O/tail.txt 2>&1
  timeout 200 python tools/dwave_latency.py 1 16 256 1024 2048 3072 4096 8192 > $O/dwave_latency.txt 2>&1
  { for k in 4 16; do timeout 200 python tools/dwave_latency.py prod $k 1 256 4096; done; timeout 200 python tools/dwave_latency.py pp 1 1024 4096
    for p in d278027-190-181 d201 d224; do echo "== $p"; DW_PARAM=$p timeout 200 python tools/dwave_latency.py 1 1024 4096; done
    echo "== f.param"; DW_PARAM=f timeout 200 python tools/dwave_latency.py 1 256 1024 2048 4096 8192; for k in 4 16; do DW_PARAM=f timeout 200 python tools/dwave_latency.py prod $k 1 256 1024; done
    echo "== g149.param"; DW_PARAM=g149 timeout 200 python tools/dwave_latency.py 1 1024 4096; DW_PARAM=g149 timeout 200 python tools/dwave_latency.py prod 4 1 256; DW_PARAM=g149 timeout 200 python tools/dwave_latency.py pp 1 1024
    echo "== a1.param / a_160_1024 / e.param (tools/agwave_latency.py)"; AG_PARAM=a1 LANE_MAX=64 timeout 300 python tools/agwave_latency.py 1 64 1024 4096; AG_PARAM=a1 LANE_MAX=1 timeout 200 python tools/agwave_latency.py prod 4 1 256; AG_PARAM=a1 LANE_MAX=64 timeout 200 python tools/agwave_latency.py pp 1 1024
    AG_PARAM=a_160_1024 LANE_MAX=64 timeout 200 python tools/agwave_latency.py 1 1024 4096; AG_PARAM=e LANE_MAX=64 timeout 200 python tools/agwave_latency.py 1 512 1024 4096; AG_PARAM=e LANE_MAX=1 timeout 200 python tools/agwave_latency.py prod 4 1 256; } > $O/small_batches.txt 2>&1
  export PBC_HIP_LIB=$R/pbc_amd/libpbc_hip.so
  [ -n "$SKIP_GLUE" ] || for p in a d159 f d201 d224 g149 a1 e; do timeout 200 oracle/_ref/glue_test pbc_amd/param/$p.param $([ $p = a1 ] && echo 30 || echo 100) latency 2>&1 | tail -n 2; [ $p = a ] || [ $p = d159 ] || continue; timeout 200 oracle/_ref/glue_test pbc_amd/param/$p.param 1048576 bench 2>&1 | tail -n 1; done > $O/glue.txt
  unset PBC_HIP_LIB; }
cd /tmp && export TMPDIR=/tmp
for w in ${PMC_WL-a d f a-prod16 d-prod16 d190 a-pp a-g1-mul f-gt-pow}; do
  B="python $R/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-host-path"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- $B > $O/kt_$w.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc1_$w -- $B > $O/pmc1_$w.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_IFETCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc2_$w -- $B > $O/pmc2_$w.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc3_$w -- $B > $O/pmc3_$w.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc4_$w -- $B > $O/pmc4_$w.log 2>&1
done
cd $R; find $O -name "*.csv" | wc -l; tail -n 1 $O/bench_a.json | cut -c1-200
