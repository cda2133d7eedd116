#!/bin/bash
# long runs of the closed loop, sampled every 50 frames: what happens around the history store's wrap / the map's spare capacity
mkdir -p gpurun_out/soak
o=gpurun_out/soak
run() {  # name frames extra-args
  timeout 600 python tools/r05_drift.py --variant ${4:-full} --frames $2 $3 --out $o/$1.jsonl > $o/$1.log 2>&1 || echo "$1 rc=$?"
  python - <<PY
import json
rows=[json.loads(l) for l in open("$o/$1.jsonl") if l.strip().startswith("{") and '"frame"' in l]
print("== $1: frame  in_use  false  used_init/new  centres_sim  raw")
for r in rows:
    if r["frame"] % 250 == 0:
        print(r["frame"], r["map_points_in_use"], r["map_points_false"], r["used_points_initial"], r["used_points_new"], round(r["centre_err_sim_max"],4), round(r["centre_err_raw_max"],3))
PY
}
run store2048 3000 "--hist-store 2048"
run store8192 5000 "--hist-store 8192"
run nochains 4500 "" no_chains
