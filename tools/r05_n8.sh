#!/bin/bash
mkdir -p gpurun_out/n8
o=gpurun_out/n8
A="--steps 40 --warmup 5 --setup-rounds 1 --no-cpu-baseline --no-secondary --no-upload-leg"
BENCH_STATE_DIGEST=1 python bench.py --gpus 1 --ba-lag 4 $A 2>$o/e1.log | tail -1 > $o/n1.json
BENCH_STATE_DIGEST=1 BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 $A 2>$o/e8.log | tail -1 > $o/n8.json
python - <<PY
import json
a=json.load(open("$o/n1.json")); b=json.load(open("$o/n8.json"))
ca, cb = a["config"], b["config"]
print("N=1 digest", ca["state_digest"], "cxx", ca["cxx_frame_loop"].get("digest"), ca["cxx_frame_loop"].get("frames_per_s"))
print("N=8 digest", cb["state_digest"], "replicas", cb["replicas"]["identical_map_records_and_poses_on_every_rank"], "lag", cb["ba_output"]["lag_key_frame_intervals"])
cx = cb["cxx_frame_loop"]
print("N=8 cxx", {k: cx.get(k) for k in ("ranks","world","cameras_per_rank","identical_digest_on_every_rank","digest","frames_per_s","transport","error","ranks_that_failed")})
print("same state", ca["state_digest"] == cb["state_digest"])
PY
tail -3 $o/e8.log | cut -c1-300
