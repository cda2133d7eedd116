"""bench.py's output contract on the MI355X: one JSON line with the driver's keys, the headline workload named, a roofline
object for the dominant kernel and (when asked for) a CPU baseline -- and the data-coupled legs really ran."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, env=None, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                         env=dict(os.environ, **dict({"BENCH_LIVE_PMC": "0"}, **(env or {}))))   # (the counter passes: one test below turns them on)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_prints_one_json_line_with_the_contract_keys(hip):
    j = _run_bench(["--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"])
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
    # the joint BA is parsed on the device from the window's key frames and its result goes back into the live map
    assert cfg["joint_ba_from_window"] is True and cfg["joint_ba_problem"]["cameras"] == 40 and cfg["joint_ba_problem"]["points"] > 200
    assert cfg["joint_ba_last"]["lm_steps"] > 0 and cfg["joint_ba_last"]["cost"] < cfg["joint_ba_last"]["cost0"]
    bo = cfg["ba_output"]
    assert bo["lag_key_frame_intervals"] == 2 and bo["windows_applied_in_timed_region"] == 2 and bo["static_points_retriangulated_last"] > 200
    assert bo["last"]["applied_at_frame"] - bo["last"]["first_key_frame"] == 30 and "pose-graph relaxation" in cfg["workload"]
    rc = cfg["register_candidates_last_frame"]
    assert cfg["intercam_last"]["lm_steps"] > 0 and rc["current_points_listed"] > 300 and rc["candidates"] > 300
    assert rc["unjudged_track_older_than_the_store"] == 0 and rc["running_verdict"]["verdicts_unjudged"] == 0
    # the rig against the synthetic truth once the gauge is taken out (a similarity of the 8 camera centres): its distortion
    assert cfg["video"]["frames"] == 120 and cfg["rig_error_vs_truth"]["centres_after_sim3_max"] < 0.06
    # the same loop from C++ through the C-ABI only (tools/cxx/frame_loop.cpp): same solves, a comparable rate
    cx = cfg["cxx_frame_loop"]
    assert "error" not in cx, cx
    assert cx["pose_ok"] is True and cx["min_live_features"] > 1500 and cx["steps"] == 10
    assert cx["joint_ba_from_window"] is True and cx["joint_cameras"] == 40
    assert cx["joint_lm_steps"] > 0 and cx["intercam_lm_steps"] > 0 and 0.5 < cx["frames_per_s"] / j["value"] < 2.0
    # ... and with every frame's images coming from pinned host memory inside the loop
    up = cfg["with_upload"]
    assert up["frames_per_s"] > 0 and 0.3 < up["ratio_to_value"] < 1.3
    # secondary rows: cfg2, cfg5, and the reference-default KLT parameter set
    assert cfg["secondary_cfg2"]["camera_frames_per_s"] > 0 and cfg["secondary_cfg5_klt"]["frames_per_s"] > 0
    rd = cfg["secondary_reference_default_klt"]
    assert rd["frames_per_s"] > 0 and min(rd["live_features"]) > 1000 and "6 levels" in rd["workload"]
    # the registration decisions: every frame's sweeps settled (single pass), and the step-for-step mode ran with every loop settled
    assert cfg["register_decision"]["frames_whose_sweeps_did_not_settle"] == 0 and cx["register_decisions_unsettled"] is False
    sq = cfg["secondary_sequential_registration"]
    assert sq["frames_per_s"] > 0 and sq["loops_whose_sweeps_did_not_settle"] == 0 and sq["ratio_to_value"] < 1.0
    r = j["roofline"]
    assert r["valu"] is not None and 0.05 < r["valu"]["frac"] < 1.0 and r["launches_per_frame"] >= 1
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["kernel"] == "k_track_rows_fused" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.005 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]


TWO_RANKS_ON_ONE_GPU = dict(BENCH_FORCE_DEVICE="0", BENCH_DIST_BACKEND="gloo")
SHORT = ["--no-cpu-baseline", "--no-secondary", "--no-cxx-loop", "--no-upload-leg"]


def test_roofline_traffic_is_measured_by_the_run_that_prints_it(hip):
    """The default bench run collects the dominant kernel's HBM counters itself (three rocprofv3 --pmc passes of the KLT stage in child
    processes: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU -- separate passes, --kernel-trace only beside them, as MI355X_MICROARCH.md asks):
    roofline.traffic says so and lands within 10 % of the committed profile; with the placed tracker it is below 1.5 x the algorithmic bytes."""
    import shutil

    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    j = _run_bench(["--gpus", "1", "--steps", "10", "--warmup", "2"] + SHORT, env=dict(BENCH_LIVE_PMC="1"))
    r = j["roofline"]
    assert r["traffic_source"].startswith("measured in this run"), r["traffic_source"]
    committed = json.load(open(os.path.join(ROOT, "profiles", "r05_tracker_pmc.json")))["traffic_bytes_per_launch"]
    assert abs(r["traffic"] - committed) < 0.1 * committed
    assert r["traffic"] < 1.5 * r["algorithmic_bytes_per_launch"]
    assert r["valu"]["source"].startswith("measured in this run") and 0.4 < r["valu"]["frac"] < 0.9


def test_bare_gpus_2_spawns_two_ranks_that_hold_one_map(hip):
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts its own two ranks (here both pinned to the one GPU of the
    box, gloo instead of RCCL, which refuses two ranks on one device).  4 cameras per rank; every frame's all-gather delivers all 8
    cameras' features and poses to both ranks, each rank replays the other's hand-back and the map update, window k is solved by
    rank k mod 2 and its packed result broadcast: after the last frame the two replicas of the map, of every camera's records and
    of the poses are bit-identical."""
    j = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--setup-rounds", "1"] + SHORT, env=TWO_RANKS_ON_ONE_GPU)
    cfg = j["config"]
    assert j["n_gpus"] == 2 and cfg["cameras_per_gpu"] == 4 and "gloo" in cfg["collectives"]["issued_by"]
    lat = cfg["collectives"]["us_per_call_measured_after_the_run"]     # every collective of the loop timed on the GPU clock
    assert "error" not in lat and lat["all_gather_features_and_poses_per_frame"] > 0 and lat["broadcast_ba_result_per_key_frame"] > 0
    assert lat["all_gather_registration_candidates_per_frame"] > 0 and lat["bytes"]["registration_candidates_per_rank"] == 3 * 4 * 4096 * 4
    assert cfg["replicas"]["ranks"] == 2 and cfg["replicas"]["identical_map_records_and_poses_on_every_rank"] is True
    assert cfg["joint_ba_from_window"] is True and cfg["joint_ba_problem"]["cameras"] == 40 and cfg["joint_ba_problem"]["measurements"] > 5000
    bo = cfg["ba_output"]
    assert bo["lag_key_frame_intervals"] == 2 and bo["windows_applied_in_timed_region"] == 4
    # rank 0 solved every other window of the timed region, and applied its own and rank 1's
    assert cfg["key_frame_solves_duty"]["joint_ba"]["solves"] == 2 and bo["last"]["solved_by_rank"] in (0, 1)
    assert cfg["intercam_last"] is None or cfg["intercam_last"]["lm_steps"] > 0
    # all 8 cameras' records are on rank 0: the pose update saw every camera
    assert len(cfg["pose_update"]["static_mapped_features_last_frame"]) == 8 and min(cfg["pose_update"]["static_mapped_features_last_frame"]) > 100


def test_bench_starts_the_cxx_loops_ranks_at_n_greater_than_one(hip):
    """`bench.py --gpus 2` with the C++ leg on: every rank starts ITS rank of tools/cxx/frame_loop.bin (north_star: host stays C++) on the
    workload file rank 0 wrote; the ranks shard the cameras like the Python loop and talk through cs_comm_* / cs_exchange_* -- here, two
    ranks on the one GPU, through the library's test transport (a real run: RCCL).  The line carries the slowest rank's rate and the
    ranks' digests agree."""
    j = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--setup-rounds", "1", "--no-cpu-baseline", "--no-secondary", "--no-upload-leg"],
                   env=TWO_RANKS_ON_ONE_GPU)
    cx = j["config"]["cxx_frame_loop"]
    assert "error" not in cx, cx
    assert cx["ranks"] == 2 and cx["world"] == 2 and cx["cameras_per_rank"] == 4 and cx["identical_digest_on_every_rank"] is True
    assert cx["transport"] == "host segment (test)" and cx["pose_ok"] is True and cx["frames_per_s"] > 50
    assert cx["apply_wait_errors"] == 0 and cx["windows_applied_in_timed_region"] == 4


def test_two_ranks_compute_what_one_rank_computes(hip):
    """The same frames, the same apply lag, one rank with 8 cameras against two ranks with 4 each: the map, every camera's records and
    the poses after the timed region are bit-identical (sha256), and so is the joint BA problem the last window parsed."""
    args = ["--steps", "20", "--warmup", "5", "--setup-rounds", "1", "--ba-lag", "2"] + SHORT
    one = _run_bench(["--gpus", "1"] + args, env=dict(BENCH_STATE_DIGEST="1"))
    two = _run_bench(["--gpus", "2"] + args, env=dict(TWO_RANKS_ON_ONE_GPU, BENCH_STATE_DIGEST="1"))
    c1, c2 = one["config"], two["config"]
    assert c1["frames_enqueued_until_end_of_timed_region"] == c2["frames_enqueued_until_end_of_timed_region"]
    assert c1["state_digest"] is not None and c1["state_digest"] == c2["state_digest"]
    assert c1["ba_output"]["windows_applied"] == c2["ba_output"]["windows_applied"] and c1["ba_output"]["last"]["window"] == c2["ba_output"]["last"]["window"]
    assert c2["replicas"]["identical_map_records_and_poses_on_every_rank"] is True
    assert c1["pose_update"]["map_points_refined"] == c2["pose_update"]["map_points_refined"]


def test_two_ranks_place_the_key_frames_where_one_rank_places_them(hip):
    """The key-frame DECISION drives the window BA (LoopConfig.keyframe_drives) and no rank waits for its device to learn it
    (--keyframe-lag 1: the decision of frame i - 1 read from pinned memory): every rank computes the decision from its replica, so two
    ranks place the same key frames, request and apply the same windows and end in the one-rank run's state, bit for bit."""
    args = ["--steps", "40", "--warmup", "5", "--setup-rounds", "1", "--ba-lag", "2", "--keyframe-drives", "1", "--keyframe-lag", "1",
            "--keyframe-ratio", "1.5"] + SHORT   # (1.5: in the bench's world the decision fires on every frame of this stretch -- a window per frame)
    one = _run_bench(["--gpus", "1"] + args, env=dict(BENCH_STATE_DIGEST="1"))
    two = _run_bench(["--gpus", "2"] + args, env=dict(TWO_RANKS_ON_ONE_GPU, BENCH_STATE_DIGEST="1"))
    c1, c2 = one["config"], two["config"]
    k1, k2 = c1["key_frame_decision"], c2["key_frame_decision"]
    n_end = c1["frames_enqueued_until_end_of_timed_region"]
    assert n_end == c2["frames_enqueued_until_end_of_timed_region"]
    p1 = [f for f in k1["key_frames_placed_by_the_decision"] if f <= n_end - 1]   # (one rank runs further legs behind the timed region: the lists are
    p2 = [f for f in k2["key_frames_placed_by_the_decision"] if f <= n_end - 1]   # compared up to its end, less the frame still in the ring)
    assert k1["decision_lag_frames"] == 1 and len(p1) >= 5 and p1 == p2, (p1[-5:], p2[-5:])
    assert c1["ba_output"]["windows_applied"] == c2["ba_output"]["windows_applied"] >= 1
    assert c1["state_digest"] is not None and c1["state_digest"] == c2["state_digest"]
    assert c2["replicas"]["identical_map_records_and_poses_on_every_rank"] is True


def test_four_ranks_compute_what_one_rank_computes(hip):
    """... and with four ranks (two cameras each, apply lag min(max(N, 2), 4) = 4, window k on rank k mod 4, the inter-camera solve of key
    frame k on rank (k + 2) mod 4): the same digest as one rank with the same lag, identical replicas."""
    args = ["--steps", "40", "--warmup", "5", "--setup-rounds", "1"] + SHORT
    one = _run_bench(["--gpus", "1", "--ba-lag", "4"] + args, env=dict(BENCH_STATE_DIGEST="1"))
    four = _run_bench(["--gpus", "4"] + args, env=dict(TWO_RANKS_ON_ONE_GPU, BENCH_STATE_DIGEST="1"))
    c1, c4 = one["config"], four["config"]
    assert four["n_gpus"] == 4 and c4["cameras_per_gpu"] == 2 and c4["ba_output"]["lag_key_frame_intervals"] == 4
    assert c4["replicas"]["ranks"] == 4 and c4["replicas"]["identical_map_records_and_poses_on_every_rank"] is True
    assert c1["state_digest"] == c4["state_digest"] and c1["ba_output"]["windows_applied"] == c4["ba_output"]["windows_applied"] > 5
    assert c1["ncc_matching"]["map_points_in_use"] == c4["ncc_matching"]["map_points_in_use"] > 7000
