# round 3, sweep 5: packed Schur waves-per-pair variants vs the wave-per-point kernels, repeated (box noise), kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s5; mkdir -p $O
run() { # name, args...
  n=$1; shift
  timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 "$@" > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} {j['value']:8.1f} frames/s")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for rep in 1 2 3; do
COSLAM_BA_PACKED=0 run unpacked_$rep
run packed_wpp2_$rep
COSLAM_HIP_LIB=$GRAFT_REPO_ROOT/coslam_amd/lib/libcoslam_hip_wpp1.so run packed_wpp1_$rep
COSLAM_HIP_LIB=$GRAFT_REPO_ROOT/coslam_amd/lib/libcoslam_hip_wpp4.so run packed_wpp4_$rep
COSLAM_BA_PACKED=0 run unpacked_noregstream_$rep --reg-stream 0
done
cd /tmp
for v in wpp2 wpp4; do
  L=""; [ $v = wpp4 ] && L=$GRAFT_REPO_ROOT/coslam_amd/lib/libcoslam_hip_wpp4.so
  rm -rf /tmp/kt_$v && COSLAM_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > /tmp/kt_$v.log 2>&1; echo "kt $v rc=$?"
  DB=$(find /tmp/kt_$v -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/ba_gaps.py $DB > $GRAFT_REPO_ROOT/$O/ba_gaps_$v.txt 2>&1; tail -14 $GRAFT_REPO_ROOT/$O/ba_gaps_$v.txt
done
