#!/bin/bash
# A/B of the direct rasteriser's threshold on the GPU box: DIRECT_LIMITS="0 32768 1000000" AB_WORKLOADS="c1_800x600_msaa4 ..." bash tools/direct_ab.sh
set -o pipefail
for limit in $DIRECT_LIMITS; do
  for w in $AB_WORKLOADS; do
    SR_DIRECT_MAX_TRIANGLES=$limit timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 100 > gpurun_out/dab_${limit}_$w.log 2>&1 || exit 1
    python - <<P
import json
d=json.loads([l for l in open("gpurun_out/dab_${limit}_$w.log") if l.startswith("{")][-1])
print("limit $limit", "$w", d["ms_per_step"], d["kernel_ms"], d["counters"]["binned"], d["counters"]["R"])
P
  done
done
