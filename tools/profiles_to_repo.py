#!/usr/bin/env python3
"""Copies the summaries of gpurun_out/profile_round into profiles/ (tracked) and derives traffic.json:
HBM bytes per launch of each kernel = (2 * FETCH_SIZE + WRITE_SIZE) KiB, the factor 2 being the gfx950
correction for FETCH_SIZE (it tallies every 128-byte line at 64 bytes, for streaming reads and for gathers alike:
tools/ubench/gather.hip; cross-checked on the vertex kernel, whose read volume is known exactly: 12 bytes per vertex)."""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
root = "gpurun_out/profile_round"
os.makedirs("profiles", exist_ok=True)
def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: take the most recent run's only"""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)
stats = newest(root + "/trace/**/*kernel_stats.csv")
shutil.copy(stats, f"profiles/{tag}_c3_kernel_stats.csv")
line = [l for l in open(root + "/trace.log") if l.startswith("{")][-1]
open(f"profiles/{tag}_c3_bench_under_rocprof.json", "w").write(line)
def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for f in [newest(path + "/**/*counter_collection.csv")]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sr::", "")
                if name.startswith(("fragment_kernel", "fragment_report_kernel")) and "true, " in name.split("<")[1]:
                    continue    # the benchmark's second figure (16x anisotropic sampler): not the launch the roofline line is about
                acc[name.split("<")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch, write = per_kernel(root + "/fetch", "FETCH_SIZE"), per_kernel(root + "/write", "WRITE_SIZE")
# (a draw of a few views starts its pass inside the vertex kernel, and a lean draw reports its counters from the shading kernel: the
# benchmark's one-view frame runs vertex_begin_kernel, setup_kernel, fragment_report_kernel and nothing else)
names = {"begin_pass_kernel": "clear", "vertex_kernel": "vertex", "vertex_begin_kernel": "vertex", "fragment_report_kernel": "fragment", "setup_kernel": "setup", "bin_kernel": "bin", "scan_kernel": "scan",
         "fill_kernel": "fill", "raster_kernel": "raster", "fragment_kernel": "fragment", "near_clip_kernel": "near_clip"}
traffic, detail = {}, {}
for k, short in names.items():
    if k in fetch or k in write:
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        traffic[short] = int((2 * f + w) * 1024)
        detail[short] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes": traffic[short]}
# the digest ties the figures to the kernel sources they were measured on: bench.py prints `traffic` only when it matches
json.dump({"kernel_source_digest": bench.kernel_source_digest(), "c3_4k_10Mtri": traffic}, open("profiles/traffic.json", "w"), indent=1)
json.dump({"workload": "c3_4k_10Mtri", "kernel_source_digest": bench.kernel_source_digest(),
           "note": "per launch averages; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB.  FETCH_SIZE tallies every 128-byte line at 64 bytes, for "
                   "streaming reads and for gathers alike (profiles/r2_fetch_size_calibration.json), hence the factor 2",
           "kernels": detail},
          open(f"profiles/{tag}_c3_hbm_traffic.json", "w"), indent=1)
print(open(f"profiles/{tag}_c3_kernel_stats.csv").read()[:1800])
print(json.dumps(detail, indent=1))
