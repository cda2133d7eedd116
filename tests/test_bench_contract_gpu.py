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
    # the same loop from C++ through the C-ABI only (tools/cxx/frame_loop.cpp): same solves, same results, a comparable rate
    cx = cfg["cxx_frame_loop"]
    assert "error" not in cx, cx
    assert cx["pose_ok"] is True and cx["min_live_features"] > 1500 and cx["steps"] == 10
    # the joint BA is data-coupled (parsed on the device from the window's key frames): the C++ loop's last solve is the one of
    # its last timed key frame, bench.py's the one behind its replays -- the same kind of problem, not the same frames
    assert cx["joint_ba_from_window"] is True and cx["joint_cameras"] == cfg["joint_ba_problem"]["cameras"] == 40
    assert 0.9 < cx["joint_measurements"] / cfg["joint_ba_problem"]["measurements"] < 1.1
    assert cx["joint_lm_steps"] > 0 and 0.8 < cx["joint_cost"] / cfg["joint_ba_last"]["cost"] < 1.25
    assert cx["intercam_lm_steps"] == cfg["intercam_last"]["lm_steps"] and 0.5 < cx["frames_per_s"] / j["value"] < 2.0
    # ... and with every frame's images coming from pinned host memory inside the loop
    up = cfg["with_upload"]
    assert up["frames_per_s"] > 0 and 0.5 < up["ratio_to_value"] < 1.2
    r = j["roofline"]
    assert r["valu"] is not None and 0.05 < r["valu"]["frac"] < 1.0 and r["launches_per_frame"] >= 1
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["kernel"] == "k_track_rows_fused" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.005 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]


def test_bench_two_ranks_on_one_gpu_feed_the_solve_from_the_gathered_records(hip):
    """The N > 1 code path on a box with ONE GPU: two ranks pinned to the same device (BENCH_FORCE_DEVICE), gloo instead of RCCL
    (which refuses two ranks on one device).  4 cameras per rank; every frame's all-gather delivers all 8 cameras' features
    and poses to both ranks, and the inter-camera solve of a key frame starts from exactly those gathered poses
    (InterCamPoseEstimator::addMapPoints, reference src/app/SL_InterCamPoseEstimator.cpp:24-37)."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BENCH_FORCE_DEVICE="0", BENCH_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                          "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    cfg = j["config"]
    assert j["n_gpus"] == 2 and cfg["cameras_per_gpu"] == 4 and "gloo" in cfg["collectives"]
    g = cfg["gathered_records"]
    # every rank saw every camera's record of the last frame, bit for bit what the owning rank packed
    assert g["cameras_checked"] == 8 and g["records_match_owner"] is True
    # ... and the inter-camera solve's initial estimate was the gathered poses (all 8), not the pre-baked ones
    assert g["intercam_start_is_gathered_pose"] is True and g["intercam_start_differs_from_prebaked"] is True
    assert cfg["intercam_last"]["lm_steps"] > 0 and cfg["intercam_last"]["cost"] < cfg["intercam_last"]["cost0"]
