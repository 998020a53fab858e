#!/bin/bash
# GPU box: the GPU test suite as it is and with every frame drawn twice, fuzzing of the newer paths against the oracle, and the bench lines of
# the workloads other than the headline one.  Everything lands under gpurun_out/round_checks/.  Each step has its own time limit and the
# script stops at the first step that fails.
set -o pipefail
out=gpurun_out/round_checks
mkdir -p $out
step() { local limit=$1 name=$2; shift 2; echo "== $name" ; timeout -k 10 $limit "$@" > $out/$name.log 2>&1 || { echo "FAILED: $name (exit $?)"; tail -30 $out/$name.log; exit 1; }; tail -2 $out/$name.log; }
step 900 tests python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider
SR_TEST_DRAWS=2 step 900 tests_two_draws python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 120 --timeout-method thread -p no:cacheprovider
SR_FUZZ_GPU=1 step 600 fuzz_default python -u tests/emu/fuzz_parity.py 20000 ${FUZZ_COUNT:-600}
SR_FUZZ_GPU=1 SR_TEST_DRAWS=2 SR_STAGED_MIN_BLOCKS=1 SR_CULL_MIN_CHUNKS=1 step 600 fuzz_two_draws_staged_culled python -u tests/emu/fuzz_parity.py 30000 ${FUZZ_COUNT:-600}
SR_FUZZ_GPU=1 SR_FUZZ_ANISOTROPIC=1 SR_TEST_DRAWS=2 step 600 fuzz_anisotropic_two_draws python -u tests/emu/fuzz_parity.py 40000 ${FUZZ_COUNT:-600}
SR_FUZZ_GPU=1 SR_FUZZ_MANY_VIEWS=1 SR_FILL_BLOCKS=1 SR_TEST_DRAWS=2 step 600 fuzz_many_views_looping python -u tests/emu/fuzz_parity.py 50000 ${FUZZ_COUNT:-600}
SR_FUZZ_GPU=1 SR_FUZZ_MANY_VIEWS=1 SR_FILL_BLOCKS=1 SR_STORE_DEPTH=1 step 600 fuzz_many_views_looping_stored_depth python -u tests/emu/fuzz_parity.py 60000 ${FUZZ_COUNT:-600}
for w in c1_800x600_msaa4 c2_1080p_64views c4_8k_10Mtri c5_50Mtri_views; do
    step 600 bench_$w python -u bench.py --workload $w
done
step 600 bench_default python -u bench.py
echo "all steps passed"
