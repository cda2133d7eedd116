#!/bin/bash
# round 6's closing measurements in one gpurun call -> gpurun_out/r06/; tools/r06_collect.py copies them into profiles/r06_*
#   1. the GPU suite and smoke
#   2. the Python loop's headline command under rocprofv3 --kernel-trace (kernel table over the timed region + a few frames' kernels per stream)
#   3. the C++ frame loop (the bench's `value`) the same way
#   4. --pmc passes (separate runs: FETCH_SIZE | WRITE_SIZE | SQ counters) of the camera group's KLT stage alone (tools/pmc_klt.py)
#   5. the bench lines: the default command and the driver's (twice)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O $O/pmc
cd $R
python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tools/r06_trace.sh r06/trace > $O/trace.log 2>&1
tools/r06_cxx_trace.sh r06/cxx_trace > $O/cxx_trace.log 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc/klt_$c -o p -- python $R/tools/pmc_klt.py > $O/pmc/klt_$c.log 2>&1
  python $R/tools/rocpd_summary.py counters $O/pmc/klt_$c/p_results.db > $O/pmc/klt_pmc_$c.md
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $O/pmc/klt_SQ -o p -- python $R/tools/pmc_klt.py > $O/pmc/klt_SQ.log 2>&1
python $R/tools/rocpd_summary.py counters $O/pmc/klt_SQ/p_results.db > $O/pmc/klt_pmc_SQ.md
rm -rf $O/pmc/klt_FETCH_SIZE $O/pmc/klt_WRITE_SIZE $O/pmc/klt_SQ
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver2.json 2> $O/bench_driver2.err
tail -3 $O/gpu_suite.log; tail -2 $O/smoke.log
python - <<'PY'
import json
for n in ("bench_default", "bench_driver", "bench_driver2"):
    try:
        d = json.loads(open(f"gpurun_out/r06/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["value_source"], d["roofline"]["frac"], d["cpu_baseline"] and d["cpu_baseline"]["value"])
    except Exception as e:
        print(n, "failed", e)
PY
