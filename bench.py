#!/usr/bin/env python
"""bench.py -- frames/sec of the track + pose + local-BA loop on MI355X (BASELINE.json metric).

One "step" = one camera frame through the hot path on each rank:
    redetect (pyramid -> KLT track -> corner detect -> top-K -> slot fill) -> advanceFrame
    -> intraCamEstimate on 192 3D-2D correspondences
    -> every BA_EVERY-th frame: local robust BA (5 key frames x 500 points, maxIter 2 / inner 10, the call the
       reference queues at src/app/SL_CoSLAM.cpp:1769)
    -> (N > 1) RCCL all-gather of {features, pose} at the inter-camera merge step.
Inputs (images, correspondences, BA problem) are resident in HBM before the timed region starts.
One camera per GPU (weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, LEVELS, FW, FH = 640, 480, 4, 50, 40
N_FEAT = FW * FH
N_FRAMES = 24
N_POSE_PTS = 192
BA_EVERY = 5
BA_KF, BA_PTS = 5, 500
HBM_PEAK_GBS = 8000.0


def klt_config():
    import coslam_amd

    # 4 levels, levelSkip 1, 7x7 window, 10 iterations/level, with gain
    return coslam_amd.KLT_SequenceTrackerConfig(nIterations=10, nLevels=LEVELS, levelSkip=1, windowWidth=7,
                                                trackWithGain=1, minCornerness=3000.0, convergenceThreshold=1.0,
                                                SSD_Threshold=20000.0, minDistance=4)


def frame_order(n):
    # ping-pong so that consecutive frames always differ by one camera step
    fwd = list(range(n))
    return fwd + fwd[-2:0:-1]


def build_inputs(cam, n_cams, seed):
    from coslam_amd.synth import Scene, make_ba_problem

    sc = Scene(n_cams, W, H, 7000, seed=seed, sigma=1.0)
    frames = np.stack([sc.render(cam, f) for f in range(N_FRAMES)])
    rng = np.random.default_rng(seed + 17 * cam)
    Ms = np.zeros((N_FRAMES, N_POSE_PTS, 3))
    ms = np.zeros((N_FRAMES, N_POSE_PTS, 2))
    R0 = np.zeros((N_FRAMES, 9))
    t0 = np.zeros((N_FRAMES, 3))
    for f in range(N_FRAMES):
        uv, vis = sc.project(cam, f)
        idx = np.nonzero(vis)[0][:N_POSE_PTS]
        Ms[f] = sc.points[idx]
        ms[f] = uv[idx] + 0.5 * rng.standard_normal((N_POSE_PTS, 2))
        ms[f, :8] += 25.0 * rng.standard_normal((8, 2))  # gross outliers for the Tukey re-weighting
        Rp, tp = sc.pose(cam, max(f - 1, 0))              # initial guess = previous frame's pose
        R0[f], t0[f] = Rp.ravel(), tp
    ba = make_ba_problem(n_cams=BA_KF, n_pts=BA_PTS, seed=seed + 99 + cam)
    return sc, frames, Ms, ms, R0, t0, ba


def cpu_baseline(frames, Ms, ms, R0, t0, K, ba, budget_s=20.0):
    """The oracle (our C restatement of the reference's path: 'port') on ONE host core."""
    import oracle

    cfg = klt_config()
    o = oracle.SequenceTracker(cfg)
    o.allocate(W, H, LEVELS, FW, FH)
    P = len(ba["pts0"])
    ptr, cam, xy, _ = oracle.csr_by_point(P, ba["obs_pt"], ba["obs_cam"], ba["obs_xy"])
    order = frame_order(N_FRAMES)
    o.detect(frames[order[0]])
    o.advanceFrame()
    t_start = time.perf_counter()
    n = 0
    while True:
        f = order[(n + 1) % len(order)]
        o.redetect(frames[f])
        o.advanceFrame()
        oracle.intracam_estimate(K, R0[f], t0[f], N_POSE_PTS, None, Ms[f], ms[f], 10.0)
        if (n + 1) % BA_EVERY == 0:
            oracle.ba_robust(ba["Ks"], ba["Rs0"], ba["ts0"], ba["pts0"], ptr, cam, xy, 2, 2, 6.0, 2, 10)
        n += 1
        if time.perf_counter() - t_start > budget_s or n >= 200:
            break
    dt = time.perf_counter() - t_start
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of the same workload (oracle/: C restatement, gcc -O2, 1 thread), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graphs", action="store_true", help="replay the KLT frame schedule from a hipGraph (default: the five launches are issued eagerly, which measures ~10 us/frame faster)")
    ap.add_argument("--no-graphs", action="store_true", help="(default behaviour; kept for older command lines)")
    ap.add_argument("--sync-ba", action="store_true", help="run the local BA on the tracking stream instead of its own")
    ap.add_argument("--no-pose", action="store_true", help="diagnostic: skip the pose leg (result not a valid bench line)")
    ap.add_argument("--serial", action="store_true", help="diagnostic: tracker, pose and BA on ONE stream (no overlap)")
    ap.add_argument("--ba-every", type=int, default=BA_EVERY, help="diagnostic: 0 disables the BA leg (result not a valid bench line)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import coslam_amd
    from coslam_amd.ba import BAWorkspace
    from coslam_amd.pose import IntraCamPoseOption, intraCamEstimate_batch_dev

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks for exercising the N > 1 code path on a box with ONE GPU: BENCH_FORCE_DEVICE pins every rank to that
    # device, BENCH_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU).  Never set by the driver.
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    dist_backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    n_gpus = max(world, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(dist_backend, rank=rank, world_size=world)

    sc, frames, Ms, ms, R0, t0, ba = build_inputs(rank, n_gpus, seed=0xC051A + 2)
    order = frame_order(N_FRAMES)

    # ---- everything resident in HBM before the clock starts -------------------------------------
    d_frames = torch.from_numpy(frames).to(dev)
    d_K = torch.from_numpy(sc.K.ravel().copy()).to(dev)
    d_Ms, d_ms = torch.from_numpy(Ms).to(dev), torch.from_numpy(ms).to(dev)
    d_R0, d_t0 = torch.from_numpy(R0).to(dev), torch.from_numpy(t0).to(dev)
    d_npts = torch.full((1,), N_POSE_PTS, dtype=torch.int32, device=dev)
    d_Ropt = torch.zeros(9, dtype=torch.float64, device=dev)
    d_topt = torch.zeros(3, dtype=torch.float64, device=dev)
    opt0 = IntraCamPoseOption()
    d_opt0 = torch.from_numpy(np.frombuffer(bytes(opt0), dtype=np.uint8).copy()).to(dev)
    d_opt = torch.zeros_like(d_opt0)
    d_ok = torch.zeros(1, dtype=torch.int32, device=dev)
    d_dest = torch.zeros(N_FEAT * 5, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(4, dtype=torch.int32, device=dev)

    # Three HIP streams, mirroring the reference's thread structure and data dependences:
    #   klt_stream   the per-frame tracker (GPUKLT::next) -- frame f+1 only needs the tracker state of frame f;
    #   pose_stream  intraCamEstimate of frame f: waits (event) for the tracker of frame f, runs next to the tracker
    #                of frame f+1; the all-gather of features||pose at the merge step (N > 1) follows it on this stream;
    #   ba_stream    local BA: the reference runs it on a worker thread next to tracking
    #                (src/app/SL_CoSLAM.cpp:1702-1730, one request in flight at a time).
    # --serial puts everything back on one stream (diagnostic).
    # The persistent tracker runs its 2000 waves in lock-step (neighbour hand-offs every pass), so a foreign wave on one
    # of its SIMDs slows the whole mesh: measured 89 -> 100-120 us per launch when pose / BA kernels share the chip.
    # The chip is therefore partitioned with CU-masked streams (hipExtStreamCreateWithCUMask): the tracker stream gets
    # all but SIDE_CUS compute units, the pose and BA streams the rest.  BENCH_SIDE_CUS=0 turns the partition off.
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    side_cus = int(os.environ.get("BENCH_SIDE_CUS", "64"))
    if args.serial or side_cus >= n_cus:
        side_cus = 0
    masked = {"ok": side_cus > 0}

    # BENCH_SIDE_MODE=range (default): the last `side_cus` CUs (whole XCDs) go to the side streams -- measured 5330
    # frames/s; =interleaved: every (n_cus/side_cus)-th CU (the tracker alone stays at 91 us, the loop drops to 4810)
    side_mode = os.environ.get("BENCH_SIDE_MODE", "range")

    def make_stream(side):
        if masked["ok"]:
            try:
                L = coslam_amd.lib()
                if side_mode == "interleaved":
                    L.cs_stream_create_cu_interleaved.restype = C.c_void_p
                    h = L.cs_stream_create_cu_interleaved(local_rank, n_cus // side_cus, 1, 0 if side else 1)
                else:
                    L.cs_stream_create_cu_range.restype = C.c_void_p
                    if side == "pose":
                        h = L.cs_stream_create_cu_range(local_rank, pose_first, pose_cus)
                    elif side:
                        h = L.cs_stream_create_cu_range(local_rank, n_cus - side_cus, side_cus - pose_from_side)
                    else:
                        h = L.cs_stream_create_cu_range(local_rank, 0, n_cus - side_cus - pose_from_klt)
                if h:
                    return torch.cuda.ExternalStream(h, device=dev)
                print("bench: CU-masked stream unavailable (" + L.cs_last_error().decode() + "); plain streams",
                      file=sys.stderr)
            except Exception as ex:  # noqa: BLE001 -- never let the partition break the benchmark
                print(f"bench: CU-masked stream unavailable ({ex}); plain streams", file=sys.stderr)
            masked["ok"] = False
        return torch.cuda.Stream(device=dev)

    pose_part = os.environ.get("BENCH_POSE_PART", "klt")
    pose_cus = int(os.environ.get("BENCH_POSE_CUS", "1"))
    pose_from_klt = pose_cus if pose_part == "own" else 0        # own: pose_cus CUs carved off the tracker's range
    pose_from_side = pose_cus if pose_part == "ownside" else 0   # ownside: carved off the BA's range
    pose_first = (n_cus - side_cus - pose_cus) if pose_part == "own" else (n_cus - pose_cus)
    klt_part = os.environ.get("BENCH_KLT_PART", "own")   # own: the complement of the side range; all: every CU
    klt_torch_stream = torch.cuda.Stream(device=dev) if klt_part == "all" else make_stream(False)
    # pose: one wave, 75 us.  It shares the tracker's partition: the tracker raises its wave priority (s_setprio), so
    # the pose wave only gets the issue slots the mesh leaves free and costs the loop nothing (6266 vs 6284 frames/s
    # without the BA leg).  BENCH_POSE_PART=klt|side|all|own|ownside (own*: BENCH_POSE_CUS CUs carved off the tracker's /
    # the BA's range -- measured slower: an uneven CU count per shader engine unbalances the tracker's mesh, and the
    # BA falls off a cliff below 64 CUs, tools/ba_partition.py)
    if args.serial:
        pose_torch_stream = klt_torch_stream
    elif pose_part == "all":
        pose_torch_stream = torch.cuda.Stream(device=dev)
    elif pose_part in ("own", "ownside") and masked["ok"]:
        pose_torch_stream = make_stream("pose")
    else:
        pose_torch_stream = make_stream(pose_part == "side")
    ba_torch_stream = klt_torch_stream if (args.sync_ba or args.serial) else make_stream(True)
    if not masked["ok"]:
        side_cus = 0
    stream = klt_torch_stream.cuda_stream
    pose_stream = pose_torch_stream.cuda_stream
    ba_stream = ba_torch_stream.cuda_stream
    # dest[] is double-buffered: the pose / exchange stage of frame f reads it while the tracker of frame f+1 writes
    d_dests = [d_dest, torch.zeros_like(d_dest)]
    klt_done = [torch.cuda.Event(), torch.cuda.Event()]
    dest_free = [torch.cuda.Event(), torch.cuda.Event()]
    trk = coslam_amd.KLT_SequenceTracker(klt_config(), device=local_rank)
    trk.allocate(W, H, LEVELS, FW, FH)
    trk.set_stream(stream)
    if side_cus > 0:
        trk.set_cu_count(n_cus if klt_part == "all" else n_cus - side_cus - pose_from_klt)
    if args.graphs and not args.no_graphs:
        trk.enable_graphs(True)
    # Frame-front prefetch (cs_klt_prefetch_dev): this frame's detector tail -- two small launches that leave the chip
    # mostly idle -- also builds the next frame's pyramid + cornerness map (horizontal fusion in the same launches).
    # (A second stream for the front was tried first: a fourth concurrently active CU-masked queue is serviced badly --
    # whichever masked stream was created last runs its kernels 3-6x longer -- 167.9 -> 178-185 us/frame.)
    prefetch = os.environ.get("BENCH_PREFETCH", "1") != "0" and not args.graphs

    P = len(ba["pts0"])
    import numpy as _np
    obs_pt = _np.asarray(ba["obs_pt"])
    o_order = _np.argsort(obs_pt, kind="stable")
    ptr = _np.zeros(P + 1, dtype=_np.int32)
    _np.add.at(ptr, obs_pt + 1, 1)
    ptr = _np.cumsum(ptr).astype(_np.int32)
    ba_ws = BAWorkspace(local_rank)
    ba_ws.upload(ba["Ks"], ba["Rs0"], ba["ts0"], ba["pts0"], ptr, ba["obs_cam"][o_order], ba["obs_xy"][o_order])
    d_baR = torch.from_numpy(ba["Rs0"].reshape(-1).copy()).to(dev)
    d_baT = torch.from_numpy(ba["ts0"].reshape(-1).copy()).to(dev)
    d_baM = torch.from_numpy(ba["pts0"].reshape(-1).copy()).to(dev)

    # all-gather payload at the merge step: features (N x 20 B) + pose (12 doubles = 96 B)
    from coslam_amd.multicam import CameraExchange
    xchg = CameraExchange(N_FEAT, dev) if world > 1 else None

    trace = [] if os.environ.get("BENCH_TRACE") else None

    def step(i):
        f = order[i % len(order)]
        b = i & 1
        if trace is not None:
            trace.append((i, "start", time.perf_counter()))
        if i >= 2 and not os.environ.get("BENCH_NO_DESTFREE"):
            klt_torch_stream.wait_event(dest_free[b])      # the consumer of this dest buffer two frames ago is done
        if prefetch:   # this frame's detector tail also builds the next frame's pyramid + cornerness map
            trk.prefetch_dev(d_frames[order[(i + 1) % len(order)]].data_ptr())
        trk.redetect_dev(d_frames[f].data_ptr(), d_dests[b].data_ptr(), d_counts.data_ptr())
        trk.advanceFrame()
        klt_done[b].record(klt_torch_stream)
        if trace is not None:
            trace.append((i, "klt", time.perf_counter()))
        pose_torch_stream.wait_event(klt_done[b])          # pose(f) consumes what the tracker produced for frame f
        with torch.cuda.stream(pose_torch_stream):
            if args.no_pose:
                pass
            else:
              d_opt.copy_(d_opt0, non_blocking=True)
              intraCamEstimate_batch_dev(pose_stream, 1, N_POSE_PTS, d_K.data_ptr(), d_R0[f].data_ptr(),
                                         d_t0[f].data_ptr(), d_npts.data_ptr(), 0, d_Ms[f].data_ptr(), d_ms[f].data_ptr(),
                                         10.0, d_Ropt.data_ptr(), d_topt.data_ptr(), d_opt.data_ptr(), d_ok.data_ptr(),
                                         device=local_rank)
            if world > 1:
                xchg.pack(d_dests[b], d_Ropt, d_topt)
                xchg.all_gather()
            dest_free[b].record(pose_torch_stream)
        if trace is not None:
            trace.append((i, "pose", time.perf_counter()))
        if args.ba_every > 0 and (i + 1) % args.ba_every == 0:
            ba_ws.solve_dev(ba_stream, d_baR.data_ptr(), d_baT.data_ptr(), d_baM.data_ptr(), 2, 2, 6.0, 2, 10)
            if trace is not None:
                trace.append((i, "ba", time.perf_counter()))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # first frame: detect (GPUKLT::first, reference src/tracking/GPUKLT.cpp:133-142)
    trk.detect_dev(d_frames[order[0]].data_ptr(), d_dests[0].data_ptr(), d_counts.data_ptr())
    trk.advanceFrame()
    for i in range(args.warmup):
        step(i + 1)
    barrier()
    t_begin = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i + 1)
    t_host = time.perf_counter() - t_begin
    barrier()
    dt = time.perf_counter() - t_begin
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    last = (args.warmup + args.steps) & 1
    if trace is not None and rank == 0:
        t00 = trace[0][2]
        for (i, what, t) in trace[-60:]:
            print(f"# step {i} {what} {1e6 * (t - t00):.1f} us", file=sys.stderr)
    n_live = int((d_dests[last].cpu().numpy().view(coslam_amd.KLT_TrackedFeature)["status"] >= 0).sum())
    pose_ok = int(d_ok.item())

    # ---- roofline of the dominant kernel: the KLT gain tracker (all levels x iterations in one persistent launch).
    # Timed with HIP events on the stream it is launched on, over a replay of the same frames right after the
    # timed region (event pairs cannot sit inside the hipGraph the timed region replays).
    roof = None
    if rank == 0:
        trk.set_profiling(True)
        n_prof = min(args.steps, 100)
        for i in range(n_prof):
            f = order[(args.warmup + args.steps + i + 1) % len(order)]
            if prefetch:
                trk.prefetch_dev(d_frames[order[(args.warmup + args.steps + i + 2) % len(order)]].data_ptr())
            trk.redetect_dev(d_frames[f].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr())
            trk.advanceFrame()
        prof = trk.get_profile()
        trk.set_profiling(False)
        hw = 7 // 2
        levels_visited = LEVELS  # levelSkip = 1
        # per feature per visited level two (2hw+2)^2 footprints of 6-byte texels, + 2 x 12 B feature I/O
        per_feature = levels_visited * 2 * (2 * hw + 2) ** 2 * 6 + 2 * 12
        alg_bytes = per_feature * N_FEAT
        launches = max(prof["launches_per_frame"], 1)
        avg_us = prof["tracker_us_total"] / max(prof["frames"], 1) / launches
        ach = (alg_bytes / launches) / (avg_us * 1e-6) / 1e9
        traffic, traffic_src = None, None
        pmc_file = os.path.join(ROOT, "profiles", "r01_tracker_pmc.json")
        if launches == 1 and os.path.exists(pmc_file):  # HBM bytes per launch from the committed rocprofv3 --pmc passes
            pj = json.load(open(pmc_file))
            traffic, traffic_src = pj["traffic_bytes_per_launch"], "profiles/r01_tracker_pmc.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)"
        roof = {"bound": "hbm", "kernel": "k_track_gain_fused" if launches == 1 else "k_track_gain_pass",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes / launches, "avg_launch_us": avg_us,
                "launches_per_frame": launches, "frames_timed": prof["frames"]}

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(frames, Ms, ms, R0, t0, sc.K, ba)

    if rank == 0:
        total_frames = args.steps * n_gpus
        out = {
            "metric": "frames/sec for track+local-BA loop (camera-frames, 640x480 x 2000 feature slots)",
            "value": total_frames / dt, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (KLT, f16 pyramid storage) + f64 (pose, BA)",
            "data": "synthetic",
            "config": {"workload": "cfg2: 1 camera/GPU 640x480, 50x40=2000 KLT slots, 4-level pyramid, 7x7 window, "
                                   "10 it/level with gain, redetect every frame; intraCamEstimate on 192 pts every "
                                   f"frame; local robust BA (5 KF x 500 pts, maxIter 2 / inner 10) every {BA_EVERY}th "
                                   "frame; all-gather of features+pose when N>1",
                       "cameras": n_gpus, "live_features_last_frame": n_live, "pose_ok": pose_ok,
                       "hip_graphs": bool(args.graphs and not args.no_graphs),
                       "frame_front_prefetch": bool(prefetch),
                       "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
                       "cu_partition": (f"tracker stream on {n_cus - side_cus} CUs, BA stream on {side_cus} CUs ({side_mode}), "
                                        f"pose stream: {pose_part}" if side_cus > 0 else "none"),
                       "streams": "one stream (--serial)" if args.serial else
                       "tracker | pose (event-ordered behind the tracker of the same frame) | local BA "
                       "(own stream, like the reference's BA worker thread)"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
