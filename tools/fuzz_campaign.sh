#!/bin/bash
# GPU box: the differential fuzzer (tests/emu/fuzz_parity.py with the shipped library) in three configurations, FUZZ_COUNT cases each
# (default 20 000), seeds from FUZZ_FIRST.  Results under gpurun_out/fuzz_campaign/.
set -o pipefail
out=gpurun_out/fuzz_campaign
mkdir -p $out
n=${FUZZ_COUNT:-20000}; f=${FUZZ_FIRST:-100000}
SR_FUZZ_GPU=1 timeout -k 10 900 python -u tests/emu/fuzz_parity.py $f $n > $out/default.log 2>&1 || { tail -5 $out/default.log; exit 1; }
tail -1 $out/default.log
SR_FUZZ_GPU=1 SR_TEST_DRAWS=2 SR_STAGED_MIN_BLOCKS=1 SR_CULL_MIN_CHUNKS=1 timeout -k 10 900 python -u tests/emu/fuzz_parity.py $((f + n)) $n > $out/two_draws_staged_culled.log 2>&1 || { tail -5 $out/two_draws_staged_culled.log; exit 1; }
tail -1 $out/two_draws_staged_culled.log
SR_FUZZ_GPU=1 SR_FUZZ_ANISOTROPIC=1 SR_TEST_DRAWS=2 timeout -k 10 900 python -u tests/emu/fuzz_parity.py $((f + 2 * n)) $n > $out/anisotropic_two_draws.log 2>&1 || { tail -5 $out/anisotropic_two_draws.log; exit 1; }
tail -1 $out/anisotropic_two_draws.log
SR_FUZZ_GPU=1 SR_FUZZ_MANY_VIEWS=1 SR_FILL_BLOCKS=1 SR_TEST_DRAWS=2 timeout -k 10 900 python -u tests/emu/fuzz_parity.py $((f + 3 * n)) $((n / 4)) > $out/many_views_looping_two_draws.log 2>&1 || { tail -5 $out/many_views_looping_two_draws.log; exit 1; }
tail -1 $out/many_views_looping_two_draws.log
