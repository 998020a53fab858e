#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages per launch."""
import csv, glob, collections, os, sys, json
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
# one pass, one directory: the newest file of each (results of earlier GPU calls stay in the local copy of the directory)
newest = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    pass_dir = os.path.relpath(f, root).split(os.sep)[0]
    if pass_dir not in newest or os.path.getmtime(f) > os.path.getmtime(newest[pass_dir]):
        newest[pass_dir] = f
for f in sorted(newest.values()):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sr::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    if not any(s in k for s in ("vertex", "setup", "scan", "fill", "raster", "fragment")):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
json.dump(out, open(root + "/summary.json", "w"), indent=1)
