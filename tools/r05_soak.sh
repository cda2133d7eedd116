#!/bin/bash
# long runs of the closed loop, sampled every 50 frames (tools/r05_drift.py): how long it sustains itself, with and without the intra-camera
# source of new map points
mkdir -p gpurun_out/soak
o=gpurun_out/soak
run() {  # name frames variant extra-args
  timeout 900 python tools/r05_drift.py --variant $3 --frames $2 $4 --out $o/$1.jsonl > $o/$1.log 2>&1 || echo "$1 rc=$?"
  python - <<PY
import json
rows=[json.loads(l) for l in open("$o/$1.jsonl") if l.strip().startswith("{")]
print("== $1: frame  in_use  false  used_init/new  centres_sim  raw  scale")
for r in rows:
    if "frame" in r and r["frame"] % 250 == 0:
        print(r["frame"], r["map_points_in_use"], r["map_points_false"], r["used_points_initial"], r["used_points_new"], round(r["centre_err_sim_max"],4), round(r["centre_err_raw_max"],3), round(r["gauge_sim"]["scale"],3))
    if "keyframe_stats" in r:
        print(json.dumps(r["keyframe_stats"])[:600])
PY
  tail -2 $o/$1.log | cut -c1-300
}
run intracam ${1:-3000} intracam ""
