#!/bin/bash
# usage: tools/exp.sh 'python-replace-snippet'  -- applies a temporary edit to sr_kernels.hip, builds, benches C3, restores
cd /root/repo
cp vulkan-renderer_amd/csrc/sr_kernels.hip /tmp/sr_kernels.exp.bak
python - "$1" "$2" <<'PY'
import sys
p='/root/repo/vulkan-renderer_amd/csrc/sr_kernels.hip'; s=open(p).read()
old,new=sys.argv[1],sys.argv[2]
assert old in s, "pattern not found"
s=s.replace(old,new)
open(p,'w').write(s)
PY
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error" ; (timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "timeout -k 10 300 python -u bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_a.log 2>&1; echo exit=\$?; tail -2 gpurun_out/bench_a.log") 2>&1 | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
cp /tmp/sr_kernels.exp.bak vulkan-renderer_amd/csrc/sr_kernels.hip; python -c "import __graft_entry__ as g; g.build()"
