set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -5 gpurun_out/gputests.log; exit 1; }
tail -1 gpurun_out/gputests.log
for t in 5 6; do for w in c2_1080p_64views c1_800x600_msaa4 c3_4k_10Mtri; do
  SR_TILE_LOG2=$t timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline > gpurun_out/tile_${t}_$w.log 2>&1 || exit 1
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/tile_${t}_$w.log") if l.startswith("{")][-1])
print($t, "$w", d["ms_per_step"], {k:d["kernel_ms"][k] for k in ("setup","bin","scan","fill","raster","fragment")})
P
done; done
