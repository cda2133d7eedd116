cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_register_gpu.py tests/test_cxx_dropin_gpu.py -x -q -m gpu 2>&1 | tail -15
