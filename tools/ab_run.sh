set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 180 --timeout-method thread > gpurun_out/ab_tests.log 2>&1 || { tail -20 gpurun_out/ab_tests.log; exit 1; }
tail -1 gpurun_out/ab_tests.log
for lib in default noprefetch; do
  for w in c5_50Mtri_views c2_1080p_64views; do
    L=""; [ $lib != default ] && L="--lib vulkan-renderer_amd/lib/exp/$lib.so"
    timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline $L > gpurun_out/ab_${lib}_$w.log 2>&1 || exit 1
    python - <<P
import json
d=json.loads([l for l in open("gpurun_out/ab_${lib}_$w.log") if l.startswith("{")][-1])
print("$lib", "$w", d["ms_per_step"], {k:d["kernel_ms"][k] for k in ("vertex","clear","setup","fragment")})
P
  done
done
