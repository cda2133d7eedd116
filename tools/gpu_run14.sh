cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02n; mkdir -p $O
for n in 0 254 252 248; do
  python bench.py --no-cpu-baseline --klt-cus $n > $O/bench_$n.json 2> $O/bench_$n.err; echo "cus $n rc=$?"; tail -c 200 $O/bench_$n.err
done
