#!/bin/bash
# A/B run on the GPU box: the shipped library and the variants under vulkan-renderer_amd/lib/exp (tools/exp_build.py, which gates every
# variant on the emulated parity suite), workloads as arguments.  AB_ARGS: extra bench.py arguments (e.g. "--steps 100").
set -o pipefail
for lib in default $(ls vulkan-renderer_amd/lib/exp 2>/dev/null | sed 's/\.so$//'); do
  for w in "$@"; do
    L=""; [ $lib != default ] && L="--lib vulkan-renderer_amd/lib/exp/$lib.so"
    timeout -k 10 250 python bench.py --workload $w --no-cpu-baseline $AB_ARGS $L > gpurun_out/ab_${lib}_$w.log 2>&1 || exit 1
    python - <<P
import json
d=json.loads([l for l in open("gpurun_out/ab_${lib}_$w.log") if l.startswith("{")][-1])
print("$lib", "$w", d["ms_per_step"], {k:d["kernel_ms"][k] for k in ("vertex","clear","setup","raster","fragment")})
P
  done
done
