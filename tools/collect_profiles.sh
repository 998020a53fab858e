#!/bin/bash
# Runs on the GPU box from the repo root: kernel trace + HBM traffic counters of the default bench run.
# Each rocprofv3 run has its own time limit; counter passes are separate runs (no trace domains mixed in).
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_round
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --headline-only > $OUT/trace.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --headline-only > $OUT/fetch.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --headline-only > $OUT/write.log 2>&1 &&
echo profiles_done
