#!/bin/bash
# window BA: the reference-compiled drop-in driver, the full GPU suite, the driver's bench command
mkdir -p gpurun_out/r03_9
./oracle/_ref/ref_ba_dropin_test > gpurun_out/r03_9/dropin.txt 2>&1; echo "dropin rc=$?"; tail -5 gpurun_out/r03_9/dropin.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03_9/pytest.txt 2>&1; tail -5 gpurun_out/r03_9/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_9/bench_driver.json 2> gpurun_out/r03_9/bench_driver.err; tail -c 3000 gpurun_out/r03_9/bench_driver.json
