cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_s12; mkdir -p $O
run() { n=$1; shift
timeout 300 python3 bench.py --no-cpu-baseline --no-secondary "$@" > $O/$n.json 2> $O/$n.err
python - $O/$n.json $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); u=j['config']['with_upload']
    print(f"{sys.argv[2]:24s} value {j['value']:8.1f}  upload {u['frames_per_s']:8.1f} ratio {u['ratio_to_value']:.3f}")
except Exception as e:
    print(sys.argv[2],'FAILED',e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
run base_300
COSLAM_STAGE_INLINE=1 run inline_300
COSLAM_STAGE_INLINE=1 run inline_20 --steps 20 --warmup 5
COSLAM_STAGE_INLINE=1 COSLAM_STAGE_BLOCKS=64 run inline64_300
