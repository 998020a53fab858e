#!/usr/bin/env python3
"""Builds a variant of the GPU library with extra compiler flags for an A/B run:
    python tools/exp_build.py NAME [-DSR_TRIS_PER_THREAD=2 ...]   ->  vulkan-renderer_amd/lib/exp/NAME.so
bench.py --lib PATH times a variant.  The variants are never loaded by anything else."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(g.PKG, "lib", "exp", name + ".so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(g.CSRC, f) for f in ("sr_kernels.hip", "sr_api.hip", "sr_assets.cpp", "sr_jpeg.cpp")]
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + g.HIPCC_FLAGS + flags + srcs + ["-lz", "-o", out])
print(out)
