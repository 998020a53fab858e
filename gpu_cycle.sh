#!/bin/bash
# usage: gpu_cycle.sh [env-for-bench-a] [env-for-bench-b]
# Builds here, then on the GPU box: GPU tests -> bench a -> bench b, chained with && so that nothing
# touches the GPU after a failing step (a failing step may be a GPU fault).  On failure the first
# failing test's report is printed, not the last lines of the log.
cd /root/repo && python -c "import __graft_entry__ as g; g.build()" || exit 1
# Gate: the same sources must first pass the gpu suite on the CPU execution harness, where an
# out-of-range access is a test failure instead of a dead GPU.
SR_TEST_EMULATED=1 timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > /tmp/emu_gate.log 2>&1 || { echo "emulated run failed: not going to the GPU"; tail -30 /tmp/emu_gate.log; exit 1; }
tail -1 /tmp/emu_gate.log
(timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "
report() { echo \"step \$1 failed\"; grep -n -m1 -A40 '^=* FAILURES\|^E  ' gpurun_out/test.log; tail -5 gpurun_out/\$2; exit 1; }
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test.log 2>&1 || report tests test.log
tail -1 gpurun_out/test.log
$1 timeout -k 10 300 python -u bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_a.log 2>&1 || report bench_a bench_a.log
tail -1 gpurun_out/bench_a.log
$2 timeout -k 10 300 python -u bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_b.log 2>&1 || report bench_b bench_b.log
tail -1 gpurun_out/bench_b.log") 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['counters']['R'], d['frame_hbm_fraction'])
    else: print(l.rstrip())
"
