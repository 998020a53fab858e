#!/usr/bin/env python3
"""Runs a micro-benchmark of this directory on the CPU first: the source is built against the tests' stand-in HIP runtime
(tests/emu: every device allocation fenced by no-access pages, kernels executed work-item by work-item) and executed once.  An
out-of-range access of any kernel ends that run with a report naming the kernel and the work-item -- instead of faulting a GPU.

    python tools/ubench/check_on_cpu.py atomics        # builds tools/ubench/_cpu/atomics and runs it; exit code 0 = no fault

No micro-benchmark goes to a GPU before this has passed for the very source that is sent."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EMU = os.path.join(ROOT, "tests", "emu")
sys.path.insert(0, EMU)
import build_emu  # noqa: E402


def main():
    name = sys.argv[1]
    out_dir = os.path.join(HERE, "_cpu")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(HERE, name + ".hip")) as fh:
        text = build_emu.translate(fh.read())
    src = os.path.join(out_dir, name + "_cpu.cpp")
    with open(src, "w") as fh:
        fh.write(f'#line 1 "{os.path.join(HERE, name + ".hip")}"\n' + text)
    exe = os.path.join(out_dir, name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-march=x86-64-v3", "-Wno-unknown-pragmas", "-U_FORTIFY_SOURCE", "-I", EMU, src,
                           os.path.join(EMU, "hipemu.cpp"), "-o", exe])
    r = subprocess.run([exe] + sys.argv[2:], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=3000)
    print(r.stdout[-4000:])
    bad = r.returncode != 0 or "memory access fault" in r.stdout or "illegal" in r.stdout.lower()
    print("FAULT OR ERROR on the CPU run: do not send this to a GPU" if bad else "cpu run clean")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
