cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_s16; mkdir -p $O
for a in "--steps 20 --warmup 5" "" "--steps 20 --warmup 5" ""; do
timeout 600 python3 bench.py --no-cpu-baseline --no-secondary $a > $O/b.json 2> $O/b.err; echo rc=$?
python - $O/b.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j['config']
    print("value", round(j['value'],1), "upload", round(c['with_upload']['frames_per_s'],1), "cxx", c['cxx_frame_loop']['frames_per_s'])
except Exception as e:
    print('FAILED',e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
