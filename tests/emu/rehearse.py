"""TEST INFRASTRUCTURE ONLY: runs bench.py (single process, small workloads) and __graft_entry__.smoke() with the host
module pointed at the CPU build of the kernels (tests/emu), to check that both scripts still work end to end with the
current library -- argument handling, counters, the JSON line, the oracle comparison.  The numbers mean nothing.
    python tests/emu/rehearse.py"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vulkan-renderer_amd"), HERE]


def main():
    import build_emu
    import softraster
    softraster.LIB_PATH = build_emu.build()
    import bench
    for workload in ("small_1080p_124ktri", "c1_800x600_msaa4", "c5_small"):
        sys.argv = ["bench.py", "--workload", workload, "--steps", "2", "--warmup", "1", "--cpu-threads", "4"]
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main()
        line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
        out = json.loads(line)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                    "config", "roofline", "cpu_baseline"):
            assert key in out, key
        assert out["config"]["workload"] == workload and out["steps"] == 2 and out["value"] > 0
        assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and out["roofline"]["bound"] == "hbm"
        assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and out["cpu_baseline"]["kind"] == "port"
        print("bench ok:", workload, out["counters"])
    import __graft_entry__
    __graft_entry__.smoke()
    print("smoke ok")


if __name__ == "__main__":
    main()
