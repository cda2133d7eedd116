#!/bin/bash
# The round's profile set (run on the GPU box from the repository root: gpurun -- tools/r05_profiles.sh); everything lands under
# gpurun_out/r05/profiles/, the summaries that are to be judged are copied into profiles/r05_* by hand afterwards.
#   1. rocprofv3 --kernel-trace of the headline command (the driver's: python bench.py, minus the secondary legs), summarised over
#      the timed region; the bench line of the traced run beside it
#   2. --pmc passes (separate runs: FETCH_SIZE | WRITE_SIZE | SQ counters) of the camera group's KLT stage alone (tools/pmc_klt.py):
#      roofline.traffic of the dominant kernel
#   3. the same three passes over a short headline loop: the pose stream's kernels (k_register_search, k_intracam, ...)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHORT="--no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg"
# 1
rocprofv3 --kernel-trace -d $O/trace -o headline -- python $R/bench.py $SHORT > $O/headline_traced_bench_line.json 2> $O/headline_trace.err
python $R/tools/rocpd_summary.py kernels $O/trace/headline_results.db --last-frames 300 > $O/headline_bench_kernel_stats.md
# 2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/klt_$c -o p -- python $R/tools/pmc_klt.py > $O/klt_$c.log 2>&1
  python $R/tools/rocpd_summary.py counters $O/klt_$c/p_results.db > $O/klt_pmc_$c.md
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $O/klt_SQ -o p -- python $R/tools/pmc_klt.py > $O/klt_SQ.log 2>&1
python $R/tools/rocpd_summary.py counters $O/klt_SQ/p_results.db > $O/klt_pmc_SQ.md
# 3
LOOP="--steps 60 --warmup 10 --setup-rounds 2 $SHORT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/loop_$c -o p -- python $R/bench.py $LOOP > $O/loop_$c.json 2> $O/loop_$c.err
  python $R/tools/rocpd_summary.py counters $O/loop_$c/p_results.db > $O/pose_stream_pmc_$c.md
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $O/loop_SQ -o p -- python $R/bench.py $LOOP > $O/loop_SQ.json 2> $O/loop_SQ.err
python $R/tools/rocpd_summary.py counters $O/loop_SQ/p_results.db > $O/pose_stream_pmc_SQ.md
rm -rf $O/trace $O/klt_FETCH_SIZE $O/klt_WRITE_SIZE $O/klt_SQ $O/loop_FETCH_SIZE $O/loop_WRITE_SIZE $O/loop_SQ
ls -la $O
# 4. the bench lines: the default command and the driver's
cd $R
python bench.py > $R/gpurun_out/r05/bench_default.json 2> $R/gpurun_out/r05/bench_default.err
python bench.py --steps 20 --warmup 5 > $R/gpurun_out/r05/bench_driver.json 2> $R/gpurun_out/r05/bench_driver.err
# 5. the C++ loop's own trace (same stretch of the sequence as the Python loop's timed region)
python - <<PY
import sys; sys.path.insert(0, "$R")
import bench
frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
sc = bench.build_scene()
bench.export_workload("/tmp/workload.bin", sc, frames, bench.build_joint_problem(sc), bench.build_ic_problem(sc), 0)
PY
export HSA_KERNARG_POOL_SIZE=$((64<<20))
cd /tmp
rocprofv3 --kernel-trace -d $O/cxx_trace -o t -- $R/tools/cxx/frame_loop.bin /tmp/workload.bin 300 30 0 2 1100 > $O/cxx_traced_line.json 2> $O/cxx_trace.err
python $R/tools/rocpd_summary.py kernels $O/cxx_trace/t_results.db --last-frames 300 > $O/cxx_frame_loop_kernel_stats.md
rm -rf $O/cxx_trace
ls -la $O $R/gpurun_out/r05
