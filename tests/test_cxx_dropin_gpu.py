"""The C++ side of the drop-in boundary, on the MI355X.

* tests/cxx/shim_link_test.cpp         the three header-compatible shims (include/shim/) with stand-in caller types;
* oracle/_ref/ref_gpuklt_dropin_test   the REFERENCE'S OWN src/tracking/GPUKLT.cpp (+ SL_Track2D.cpp and the data model)
                                       compiled in place over the shims, driven as SingleSLAM drives it, and checked slot
                                       by slot against the on-device hand-back (cs_klt_handback_dev);
* oracle/_ref/ref_lazy_adaptor_test    the same sources + SL_SingleSLAM.cpp under the lazy list adaptor (include/shim/tracking/GPUKLTGroup.h).
* oracle/_ref/ref_ba_dropin_test       the REFERENCE'S OWN src/app/SL_CoSLAMRobustBA.cpp (parseInputs / run),
                                       SL_InterCamPoseEstimator.cpp (addMapPoints / apply) and SL_SingleSLAM.cpp
                                       (chooseStaticFeatPts ...) compiled in place over the shims.
* oracle/_ref/ref_register_test        the REFERENCE'S OWN searchMahaNearestFeatPt (src/app/SL_SingleSLAM.cpp:1141-1164) over
                                       FeaturePoints lists built with the reference's classes, against cs_register_search.
* oracle/_ref/ref_ncc_test             the REFERENCE'S OWN NCCBlock::computeScaled, matchNCCBlock (src/slam/SL_NCCBlock.cpp) and
                                       getEpiNccMat (src/slam/SL_FeatureMatching.cpp) against cs_ncc_match_between.
* oracle/_ref/ref_posegraph_test       the REFERENCE'S OWN GlobalPoseGraph::computeNewCameraRotations / computeNewCameraTranslations
                                       (src/slam/SL_GlobalPoseEstimation.cpp) against relaxPoseGraphs / cs_posegraph_* on graphs
                                       built with the reference's classes; ref_posegraph_methods_test: the two member functions
                                       taken from include/shim/slam/coslam_posegraph.h instead, checked against the golden file
                                       the first binary writes on the CPU.
* oracle/_ref/ref_export_test          the REFERENCE'S OWN CoSLAM::exportResultsVer1 (src/app/SL_CoSLAM.cpp) against
                                       cs_export_results_v1: six text files, byte for byte.
The oracle/_ref binaries are built by oracle/Makefile where the reference tree exists (__graft_entry__.build()) and
travel with the repo snapshot; the reference sources themselves are never copied."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(exe, ok_text):
    assert os.path.exists(exe), f"{exe} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` where /root/reference exists"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert ok_text in out.stdout, out.stdout + out.stderr
    return out.stdout


def test_reference_gpuklt_source_runs_over_the_shim_and_matches_the_device_handback(hip):
    out = _run(os.path.join(ROOT, "oracle", "_ref", "ref_gpuklt_dropin_test"), "ref GPUKLT drop-in ok")
    print(out)


def test_lazy_feature_list_adaptor_leaves_the_lists_as_the_reference_would(hip):
    """SURVEY 8f-1, second half (include/shim/tracking/GPUKLTGroup.h): three cameras' frames run as a device-resident group, the reference's
    FeaturePoints / Track2D lists rebuilt only at frames 4, 9 and 13 -- tracks, frame lists and the reference's own
    SingleSLAM::chooseStaticFeatPts / getNumMappedStaticPts equal the synchronous drop-in's"""
    out = _run(os.path.join(ROOT, "oracle", "_ref", "ref_lazy_adaptor_test"), "lazy adaptor ok")
    print(out)


def test_reference_ba_callers_run_over_the_shim(hip):
    out = _run(os.path.join(ROOT, "oracle", "_ref", "ref_ba_dropin_test"), "ref BA callers drop-in ok")
    assert "RobustBundleRTS drop-in ok" in out and "InterCamPoseEstimator drop-in ok" in out
    print(out)


def test_reference_search_function_agrees_with_the_registration_kernel(hip):
    out = _run(os.path.join(ROOT, "oracle", "_ref", "ref_register_test"), "candidates agree with the reference's searchMahaNearestFeatPt")
    print(out)


def test_reference_ncc_code_agrees_with_the_ncc_kernels(hip):
    out = _run(os.path.join(ROOT, "oracle", "_ref", "ref_ncc_test"), "equal the reference's bit for bit")
    print(out)


def test_reference_posegraph_code_agrees_with_the_relaxation_kernel(hip, tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_posegraph_test")
    out = _run(exe, "vs the reference's own methods")
    print(out)
    gold = str(tmp_path / "pg.bin")
    subprocess.run([exe, "golden", gold], check=True, timeout=600)        # CPU: the reference's methods
    m = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_posegraph_methods_test"), gold], capture_output=True, text=True,
                       timeout=600)
    assert m.returncode == 0 and "ref_posegraph_methods_test: OK" in m.stdout, m.stdout + m.stderr
    print(m.stdout)


def test_reference_export_code_agrees_with_the_result_writer(hip, tmp_path):
    """the reference's own CoSLAM::exportResults (src/app/SL_CoSLAM.cpp compiled in place) against cs_export_results_v1; host
    code, also run by the CPU suite (tests/test_results_cpu.py)"""
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_export_test"), str(tmp_path)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0 and "ref_export_test: OK" in out.stdout, out.stdout + out.stderr


def test_cxx_shims_link_and_run(hip):
    """include/shim/: V3D_GPU::KLT_SequenceTracker, intraCamEstimate, bundleAdjustRobust with stand-in caller types."""
    src = os.path.join(ROOT, "tests", "cxx", "shim_link_test.cpp")
    exe = os.path.join(ROOT, "tests", "cxx", "shim_link_test.bin")
    libdir = os.path.join(ROOT, "coslam_amd", "lib")
    cmd = ["g++", "-std=c++11", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "include", "shim"), src,
           "-L", libdir, "-lcoslam_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim ok" in out.stdout


def test_cxx_frame_loop_on_two_ranks_ends_in_the_one_rank_runs_map(hip, tmp_path):
    """tools/cxx/frame_loop.cpp at N > 1 (VERDICT r04 missing 4: "host stays C++"): rank r owns cameras r * nc ..; per frame ONE all-gather
    of {dest[], R, t} through cs_exchange_*, the other ranks' cameras replayed through the same hand-back, the registration candidates'
    and the NCC records' all-gathers, key-frame window k solved by rank k % N and its record broadcast (cs_comm_*).  RCCL refuses two
    ranks on one device, so the two ranks here share the GPU and talk through the library's TEST transport (cs_comm_create_host: the same
    entry points staged through a shared-memory segment -- what gloo is for the Python loop's tests); the launch-per-pass tracker, because
    two persistent trackers of two processes cannot promise co-residency.  Both ranks must end in the SAME map / tables / poses (an
    FNV-1a digest) as the one-rank run of the same frames: every collective carried what the one-rank loop reads in place."""
    import json
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "cxx", "frame_loop.bin")
    assert os.path.exists(exe), "tools/cxx/frame_loop.bin missing: __graft_entry__.build()"
    sys.path.insert(0, root)
    import bench

    wl = str(tmp_path / "workload.bin")
    frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
    sc = bench.build_scene()
    bench.export_workload(wl, sc, frames, bench.build_joint_problem(sc), bench.build_ic_problem(sc), 0)
    del frames
    steps, warm = "60", "10"
    base = dict(os.environ, COSLAM_KLT_FUSED="0", HSA_KERNARG_POOL_SIZE=str(64 << 20))

    def line(out):
        return json.loads([x for x in out.splitlines() if x.startswith("{")][-1])

    one = subprocess.run([exe, wl, steps, warm, "0", "2"], env=base, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = line(one.stdout)
    assert j1["world"] == 1 and j1["pose_ok"] and j1["windows_applied_in_timed_region"] >= 10
    # the registration's launches fused (the second visits' lists built by the walks, advance + refine as one launch: the default) against
    # the launch-per-step sequence: the same map, tables and poses, the same second visits
    plain = subprocess.run([exe, wl, steps, warm, "0", "2"], env=dict(base, COSLAM_FUSED_ROUNDS="0"), capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    jp = line(plain.stdout)
    assert jp["digest"] == j1["digest"], "the fused registration launches do not end where the launch-per-step sequence does"
    for k in ("second_visit_features_attached", "second_visit_conflicts", "map_points_in_use", "second_visit_points_beyond_the_list"):
        assert jp[k] == j1[k], (k, jp[k], j1[k])
    assert j1["second_visit_features_attached"] > 0
    seg = f"/coslam_cxx_{os.getpid()}"
    procs = [subprocess.Popen([exe, wl, steps, warm, "0", "2"], env=dict(base, RANK=str(r), WORLD_SIZE="2", COSLAM_FORCE_DEVICE="0",
                                                                        COSLAM_COMM="host:" + seg),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        outs.append(line(o))
    assert [o["rank"] for o in outs] == [0, 1] and all(o["world"] == 2 and o["cameras_per_rank"] == 4 for o in outs)
    assert all(o["pose_ok"] and o["apply_wait_errors"] == 0 and not o["register_decisions_unsettled"] for o in outs)
    assert outs[0]["digest"] == outs[1]["digest"], "the two ranks' replicas differ"
    assert outs[0]["digest"] == j1["digest"], "two ranks do not end where one rank does"
    assert outs[0]["map_points_in_use"] == j1["map_points_in_use"] > j1["map_points_at_start"]
    # ---- the key frames where the DECISION puts them, no host wait per frame (COSLAM_KEYFRAME_DRIVES=1, lag 1: VERDICT r05 item 6's C++
    # counterpart): the C++ loop places the key frames the Python loop places on the same frames, requests and applies their windows, and two
    # C++ ranks -- each reading its own replica's decision -- end in the one-rank run's state
    kf_env = dict(base, COSLAM_KEYFRAME_DRIVES="1", COSLAM_KEYFRAME_LAG="1", COSLAM_KEYFRAME_RATIO="1.5")
    kd = subprocess.run([exe, wl, steps, warm, "0", "2"], env=kf_env, capture_output=True, text=True, timeout=600)
    assert kd.returncode == 0, kd.stderr[-2000:]
    jk = line(kd.stdout)
    placed = jk["key_frames_placed_by_the_decision"]
    assert jk["keyframe_lag"] == 1 and jk["pose_ok"] and len(placed) >= 10 and jk["windows_applied"] >= 3 and jk["apply_wait_errors"] == 0
    assert jk["windows_not_applied_history_too_short"] == 0 and jk["digest"] != j1["digest"]      # (another schedule of windows: another map)
    seg2 = f"/coslam_cxxkf_{os.getpid()}"
    procs = [subprocess.Popen([exe, wl, steps, warm, "0", "2"], env=dict(kf_env, RANK=str(r), WORLD_SIZE="2", COSLAM_FORCE_DEVICE="0", COSLAM_COMM="host:" + seg2),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs2 = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        outs2.append(line(o))
    for o in outs2:
        assert o["key_frames_placed_by_the_decision"] == placed, (o["rank"], len(o["key_frames_placed_by_the_decision"]), len(placed),
                                                                  [a for a in placed if a not in o["key_frames_placed_by_the_decision"]][:5])
        assert o["windows_applied"] == jk["windows_applied"] and o["apply_wait_errors"] == 0, (o["rank"], o["windows_applied"], jk["windows_applied"],
                                                                                               o["apply_wait_errors"])
    assert outs2[0]["digest"] == outs2[1]["digest"] == jk["digest"], "two ranks placing their key frames do not end where one rank does"
    # the Python loop over the same frames: the same decisions
    import torch

    from coslam_amd.frameloop import FrameLoop, LoopConfig

    dev = torch.device("cuda", 0)
    NA = bench.N_CAMS
    frames = bench.render_video(list(range(NA)), bench.N_FRAMES)
    video = {c: torch.from_numpy(frames[c]).to(dev) for c in range(NA)}
    cfg = LoopConfig(n_cams=NA, W=bench.W, H=bench.H, levels=bench.LEVELS, fw=bench.FW, fh=bench.FH, pts_stride=bench.PTS_STRIDE, n_col_blk=bench.N_COL_BLK,
                     n_row_blk=bench.N_ROW_BLK, key_every=bench.KEY_EVERY, p_reg=bench.P_REG, keyframe_drives=True, keyframe_ratio=1.5, keyframe_lag=1)
    lp = FrameLoop(cfg, bench.build_scene(), video, None, bench.klt_config(), bench.reg_covariances(len(sc.points)), rank=0, world=1, device=0,
                   associate=bench.associate)
    lp.first_frame()
    n_frames = jk["frames_run"]   # (set-up intervals + warm-up + steps)
    assert n_frames >= int(steps) + int(warm)
    for i in range(1, n_frames + 1):
        lp.step(i, False)
    lp.drain()
    py_placed = lp.keyframe_stats()["key_frames_placed_by_the_decision"]
    assert py_placed == placed, (py_placed[:10], placed[:10], len(py_placed), len(placed))
    assert lp.applied == jk["windows_applied"]
