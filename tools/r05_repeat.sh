mkdir -p gpurun_out/rep
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/rep/d$i.json 2> gpurun_out/rep/d$i.err; echo "driver $i rc=$?"
done
for i in 1 2; do
  python bench.py > gpurun_out/rep/f$i.json 2> gpurun_out/rep/f$i.err; echo "default $i rc=$?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/rep/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); c = j["config"]
    print(f, round(j["value"],1), "cxx", round(c["cxx_frame_loop"].get("frames_per_s", -1),1), "unsettled", c["register_decision"]["frames_whose_sweeps_did_not_settle"], "wait_err", c["ba_output"]["apply_wait_errors"], "rig", round(c["rig_error_vs_truth"]["centres_after_sim3_max"],4), "pose_ok", all(c["pose_ok"]))
PY
