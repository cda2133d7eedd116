#!/bin/bash
mkdir -p gpurun_out/r03_22
COSLAM_BA_SEGTIME=1 timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop --only-solve joint > gpurun_out/r03_22/joint.json 2> gpurun_out/r03_22/joint.err
grep "ba segtime" gpurun_out/r03_22/joint.err | sed -n 8,20p
COSLAM_BA_SEGTIME=1 COSLAM_BA_WINDOW_CHUNK=1 timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop --only-solve joint > gpurun_out/r03_22/joint1.json 2> gpurun_out/r03_22/joint1.err
grep "ba segtime" gpurun_out/r03_22/joint1.err | sed -n 8,14p
