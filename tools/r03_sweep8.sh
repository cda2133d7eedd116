cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s8; mkdir -p $O
timeout 900 python -m pytest tests/test_pose_ba_gpu.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
run() { # name, args...
  n=$1; shift
  timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 "$@" > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} {j['value']:8.1f} frames/s")
except Exception as e:
    print(sys.argv[2], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
for rep in 1 2 3; do
COSLAM_BA_PACKED=0 run unpacked_$rep
run packed_$rep
COSLAM_BA_PACKED=0 run unpacked_cams4_$rep --klt-cams-per-launch 4
run packed_cams4_$rep --klt-cams-per-launch 4
COSLAM_BA_PACKED=0 run unpacked_cams3_$rep --klt-cams-per-launch 3
COSLAM_BA_PACKED=0 run unpacked_cams5_$rep --klt-cams-per-launch 5
done
