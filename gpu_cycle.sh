#!/bin/bash
# usage: gpu_cycle.sh [extra env]   -- builds, runs GPU tests + C3 bench on the GPU box, prints summary
cd /root/repo && python -c "import __graft_entry__ as g; g.build()" || exit 1
(timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test.log 2>&1; echo tests_exit=\$?; tail -3 gpurun_out/test.log; $1 timeout -k 10 300 python -u bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_a.log 2>&1; echo exit=\$?; tail -2 gpurun_out/bench_a.log; $2 timeout -k 10 300 python -u bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_b.log 2>&1; echo exit=\$?; tail -2 gpurun_out/bench_b.log") 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['counters']['R'], d['frame_hbm_fraction'])
    else: print(l.rstrip())
"
