#!/bin/bash
# Runs on the GPU box from the repo root: FETCH_SIZE of tools/ubench/gather (built beforehand with
# hipcc --offload-arch=gfx950 -O3 tools/ubench/gather.hip -o tools/ubench/gather) against read volumes known by construction.
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fetch_calibration
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout -s KILL 60 $GRAFT_REPO_ROOT/tools/ubench/gather > $OUT/plain.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $GRAFT_REPO_ROOT/tools/ubench/gather > $OUT/fetch.log 2>&1 &&
echo calibration_done
