cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_s18; mkdir -p $O
for a in "--steps 20 --warmup 5" "" "--ba-prebaked" "--steps 20 --warmup 5 --ba-prebaked"; do
timeout 600 python3 bench.py --no-cpu-baseline --no-secondary --no-cxx-loop $a > $O/b.json 2> $O/b.err; echo rc=$?
python - $O/b.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j['config']
    print("value", round(j['value'],1), "upload", round(c['with_upload']['frames_per_s'],1), c['joint_ba_problem'], c['joint_ba_last'], "pose_err", c['pose_translation_error_vs_truth'], c['pose_correspondences'])
except Exception as e:
    print('FAILED',e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
