#!/bin/bash
# usage: tools/ab_variants.sh ["variantA" "variantB" ...]
# A/B of pipeline variants (SR_VARIANT, see sr_api.hip) on the GPU box, one gpurun call:
#   for every variant: GPU parity tests, then the C3 benchmark; stops at the first failing step.
# Every variant must first pass the gpu suite on the CPU execution harness here (tests/emu), so that an
# indexing mistake shows up as a failed test on the CPU and not as a GPU fault.
cd /root/repo || exit 1
VARIANTS=("$@")
if [ ${#VARIANTS[@]} -eq 0 ]; then
    VARIANTS=("default" "fused_geometry" "cluster_cull" "dirty_blocks" "cluster_cull,dirty_blocks" "cluster_cull,sparse_tail,dirty_blocks" "cluster_cull,sparse_tail,dirty_blocks,exact_shading" "cluster_cull,sparse_tail,dirty_blocks,near_clip" "near_clip")
fi
python -c "import __graft_entry__ as g; g.build()" || exit 1
for v in "${VARIANTS[@]}"; do
    vv=$v; [ "$v" = "default" ] && vv=""
    SR_VARIANT=$vv SR_TEST_EMULATED=1 timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > /tmp/emu_gate.log 2>&1 \
        || { echo "variant '$v' fails on the CPU execution harness: not going to the GPU"; tail -30 /tmp/emu_gate.log; exit 1; }
    echo "emulated ok: $v ($(tail -1 /tmp/emu_gate.log))"
done
CMD=""
for v in "${VARIANTS[@]}"; do
    vv=$v; [ "$v" = "default" ] && vv=""
    tag=$(echo "$v" | tr ',' '+')
    CMD+="echo '== $v'; SR_VARIANT='$vv' timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test_$tag.log 2>&1 || { grep -n -m1 -A30 '^=* FAILURES' gpurun_out/test_$tag.log; exit 1; }; tail -1 gpurun_out/test_$tag.log; "
    for wl in c3_4k_10Mtri c1_800x600_msaa4 c2_1080p_64views; do   # the headline, then two scenes of large triangles (what a variant must not slow down)
        CMD+="SR_VARIANT='$vv' timeout -k 10 300 python -u bench.py --workload $wl --steps 30 --no-cpu-baseline > gpurun_out/bench_${tag}_$wl.log 2>&1 || { tail -5 gpurun_out/bench_${tag}_$wl.log; exit 1; }; tail -1 gpurun_out/bench_${tag}_$wl.log; "
    done
done
(timeout 1500 /usr/local/graft/bin/gpurun --timeout 1100 -- "$CMD") 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('   %-20s %-40s %8.1f Mtris/s  %.4f ms  hbm %.3f' % (d['config']['workload'], d['config'].get('pipeline_variant'), d['value'], d['ms_per_step'], d['frame_hbm_fraction']))
        print('   ', d['kernel_ms'])
    else:
        print(l.rstrip())
"
