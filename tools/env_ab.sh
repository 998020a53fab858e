#!/bin/bash
# GPU box: the same library with and without an environment switch.  AB_ENV: NAME=VALUE set for the "off" runs; workloads as arguments.
set -o pipefail
out=gpurun_out/env_ab
mkdir -p $out
for w in "$@"; do
    for mode in on off on off; do
        if [ $mode = off ]; then export "$AB_ENV"; else unset "${AB_ENV%%=*}"; fi
        timeout -k 10 300 python -u bench.py --workload $w --headline-only > $out/${w}_$mode.log 2>&1 || { tail -20 $out/${w}_$mode.log; exit 1; }
        python - $out/${w}_$mode.log $w $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "ms_per_step", d["ms_per_step"], {k: v for k, v in d["kernel_ms"].items() if k in ("vertex", "setup", "raster", "fragment")})
PY
    done
done
