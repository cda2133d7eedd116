"""bench.py's output contract on the MI355X: one JSON line with the driver's keys, the headline workload named, a roofline
object for the dominant kernel and (when asked for) a CPU baseline -- and the data-coupled legs really ran."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys(hip):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 10 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["unit"] == "frames/s" and j["vs_baseline"] is None and j["data"] == "synthetic" and j["scaling"] == "strong"
    assert abs(j["value"] - 1e3 / j["ms_per_step"]) < 1e-6 * j["value"]
    cfg = j["config"]
    assert "8 cams" in cfg["workload"] and "joint local BA" in cfg["workload"] and "inter-camera" in cfg["workload"]
    assert cfg["cameras"] == 8 and all(cfg["pose_ok"]) and min(cfg["pose_correspondences"]) > 50
    assert min(cfg["live_features_last_frame"]) > 1500
    assert cfg["joint_ba_last"]["lm_steps"] > 0 and cfg["joint_ba_last"]["cost"] < cfg["joint_ba_last"]["cost0"]
    assert cfg["posegraph_last"]["nodes"] == 8 * 21 and cfg["posegraph_last"]["components"] == 32
    assert cfg["posegraph_last"]["max_non_key_translation_change"] > 1e-4 and "pose-graph relaxation" in cfg["workload"]
    assert cfg["intercam_last"]["lm_steps"] > 0 and cfg["register_candidates_last_frame"]["current_static"] > 1000
    r = j["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["kernel"] == "k_track_rows_fused" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.01 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
