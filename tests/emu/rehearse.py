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
        assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_utilisation"} and out["roofline"]["bound"] == "hbm"
        if out["config"]["views_per_gpu"] == 1:     # the moving-camera companion of a one-view workload
            assert out["ms_per_step_orbit"] > 0 and out["ms_per_step_camera_jumps"] > 0
            assert out["moving_camera"]["orbit"]["draws"] == 2 and out["moving_camera"]["jumps"]["draws"] == 2
        assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and out["cpu_baseline"]["kind"] == "port"
        print("bench ok:", workload, out["counters"])
    # the multi-rank paths (what the round-end driver runs on a whole node), two ranks over gloo on the CPU build
    import subprocess
    for extra in (["--workload", "c5_small"], ["--mode", "sortfirst", "--workload", "small_1080p_124ktri"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--lib", softraster.LIB_PATH] + extra,
                           env=dict(os.environ, SR_BENCH_CPU_REHEARSAL="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["n_gpus"] == 2 and out["value"] > 0 and out["n1_same_workload_Mtris_per_s"] > 0
        assert out["speedup_vs_n1_same_workload"] > 0 and out["n1_same_workload_ms_per_step"] > 0
        if out["value"] > 100:     # (both are rounded in the line: comparable only where the rates are not tiny)
            ratio = out["value"] / out["n1_same_workload_Mtris_per_s"]
            assert abs(out["speedup_vs_n1_same_workload"] - ratio) < 0.05 * ratio, (out["speedup_vs_n1_same_workload"], ratio)
        print("bench ok: two ranks,", " ".join(extra), "speed-up over one rank on the same workload", out["speedup_vs_n1_same_workload"])
    import __graft_entry__
    __graft_entry__.smoke()
    print("smoke ok")


if __name__ == "__main__":
    main()
