cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
timeout 600 python -m pytest tests/test_register_gpu.py tests/test_cxx_dropin_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/reg_time.py 2>&1 | tail -2
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo bench rc=$?; tail -c 300 $O/bench.err
