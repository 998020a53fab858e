#!/usr/bin/env python3
"""Per-kernel fingerprint of the gfx950 machine code of vulkan-renderer_amd/csrc/sr_kernels.hip.

    tools/isa_check.py --save profiles/tested_isa.json     record the fingerprints of the current sources
    tools/isa_check.py --compare profiles/tested_isa.json  list the kernels whose code differs from the record

Use: a kernel whose fingerprint equals the one recorded when the GPU tests last passed is the same
machine code, whatever was edited around it -- it needs no new GPU run to be trusted.  A kernel that
differs has not been on a GPU yet.
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vulkan-renderer_amd", "csrc", "sr_kernels.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-fvisibility=hidden"]


def device_asm() -> str:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["hipcc", *FLAGS, "--cuda-device-only", "-S", "-o", out, SRC], stderr=subprocess.DEVNULL)
        return open(out).read()


def fingerprints(asm: str) -> dict:
    """symbol -> {sha, lines, vgprs, sgprs, lds, scratch}"""
    out = {}
    body = re.compile(r"^(_Z\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:", re.S | re.M)
    for m in body.finditer(asm):
        name, text = m.group(1), m.group(2)
        code = "\n".join(l.split(";")[0].rstrip() for l in text.splitlines() if l.strip() and not l.lstrip().startswith(";"))
        code = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", code)   # local labels carry the function's ordinal in the file
        # instructions and labels only: directives (kernel descriptor, section names, kernarg size) name the symbol
        code = "\n".join(l for l in code.splitlines() if not l.lstrip().startswith(".") or l.lstrip().startswith(".LBB_"))
        code = re.sub(r"_Z\w+", "SYM", code)
        out[name] = {"sha": hashlib.sha256(code.encode()).hexdigest()[:16], "lines": code.count("\n") + 1}
    for m in re.finditer(r"\.amdhsa_kernel (\w+)\n(.*?)\.end_amdhsa_kernel", asm, re.S):
        k = out.get(m.group(1))
        if k is None:
            continue
        for key, pat in (("vgprs", r"\.amdhsa_next_free_vgpr (\d+)"), ("sgprs", r"\.amdhsa_next_free_sgpr (\d+)"),
                         ("lds", r"\.amdhsa_group_segment_fixed_size (\d+)"), ("scratch", r"\.amdhsa_private_segment_fixed_size (\d+)")):
            mm = re.search(pat, m.group(2))
            k[key] = int(mm.group(1)) if mm else None
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt", *names], stdout=subprocess.PIPE, text=True, check=True)
        return dict(zip(names, r.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save")
    ap.add_argument("--compare")
    a = ap.parse_args()
    fp = fingerprints(device_asm())
    pretty = demangle(list(fp))
    if a.save:
        with open(a.save, "w") as fh:
            json.dump({pretty[k]: v for k, v in sorted(fp.items())}, fh, indent=1)
        print(f"{len(fp)} kernels recorded in {a.save}")
    if a.compare:
        ref = json.load(open(a.compare))
        cur = {pretty[k]: v for k, v in fp.items()}
        changed = [k for k in cur if k in ref and cur[k]["sha"] != ref[k]["sha"]]
        new = [k for k in cur if k not in ref]
        gone = [k for k in ref if k not in cur]
        # a kernel that only changed its name (say, a template parameter was added) but not its code
        ref_by_sha = {ref[k]["sha"]: k for k in gone}
        renamed = [(k, ref_by_sha[cur[k]["sha"]]) for k in new if cur[k]["sha"] in ref_by_sha]
        for k, was in renamed:
            print(f"RENAMED  {k}\n         same code as recorded {was}")
            new.remove(k)
            gone.remove(was)
        for tag, lst in (("CHANGED", changed), ("NEW", new), ("REMOVED", gone)):
            for k in lst:
                v = cur.get(k) or ref[k]
                print(f"{tag:8s} {k}  vgprs={v.get('vgprs')} lds={v.get('lds')} scratch={v.get('scratch')}")
        print(f"{len(cur) - len(changed) - len(new)} of {len(cur)} kernels are the recorded machine code; {len(new)} new, {len(changed)} changed, {len(gone)} removed")
        sys.exit(1 if (changed or gone) else 0)
    if not a.save and not a.compare:
        for k, v in sorted(fp.items()):
            print(f"{v['sha']} {v['lines']:6d} lines vgprs={v.get('vgprs')} sgprs={v.get('sgprs')} lds={v.get('lds')} scratch={v.get('scratch')}  {pretty[k]}")


if __name__ == "__main__":
    main()
