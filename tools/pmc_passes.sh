#!/bin/bash
# Runs on the GPU box (from the repo root): one rocprofv3 counter pass per run, each under its own
# time limit, chained with &&.  Results: gpurun_out/pmc/<pass>/.../*counter_collection.csv
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${PMC_OUT:-pmc}   # PMC_BENCH_ARGS="--workload c5_50Mtri_views" PMC_OUT=pmc_c5: another workload
rm -rf $OUT; mkdir -p $OUT
cd /tmp
run() { # name counters...
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --headline-only $PMC_BENCH_ARGS > $OUT/$name.log 2>&1
}
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU &&
run sq_b SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS &&
run sq_c SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 &&
run grbm GRBM_GUI_ACTIVE GRBM_COUNT &&
run ta TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum &&
echo pmc_done
