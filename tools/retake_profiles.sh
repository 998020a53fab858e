#!/bin/bash
# After a GPU call that ran `bash tools/collect_profiles.sh && bash tools/pmc_passes.sh`: copies the summaries into profiles/ (tag r3).
set -e
cd "$(dirname "$0")/.."
python tools/profiles_to_repo.py ${1:-r3} > /dev/null
python tools/pmc_summary.py > /dev/null
python - <<'P'
import json, sys
sys.path.insert(0, '.')
import bench
old = json.load(open("profiles/r2_c3_pmc.json"))
new = json.load(open('gpurun_out/pmc/summary.json'))
old['kernels'] = {k: v for k, v in new.items() if k.startswith(('vertex_begin_kernel', 'setup_kernel<1, false, true', 'fragment_report_kernel<1, false, true>'))}
old['kernel_source_digest'] = bench.kernel_source_digest()
json.dump(old, open('profiles/r3_c3_pmc.json', 'w'), indent=1)
print("profiles retaken for kernel sources", old['kernel_source_digest'])
P
