#!/bin/bash
# Collects PMC counters for bench.py in separate rocprofv3 passes (one --pmc set per run).
# Usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <tag> [bench args...]   (bench.py --no-cpu <args>: with
# no arguments the command whose line is committed, without the leg that keeps two frames in flight -- overlapping
# frames stretch each other's kernel durations -- so that launch times can be compared)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_pass () {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-overlap-leg $BENCH_ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
BENCH_ARGS="$*"
run_pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run_pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_THREAD_CYCLES_VALU
run_pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run_pass tcc1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
run_pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
# kernel durations of the same command (no counters)
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-overlap-leg $BENCH_ARGS > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
echo passes done
