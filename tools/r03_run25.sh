#!/bin/bash
O=gpurun_out/r03_25; mkdir -p $O
COSLAM_BA_SEGTIME=1 timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop --only-solve joint > $O/joint.json 2> $O/joint.err
grep "ba segtime" $O/joint.err | sed -n 30,42p
COSLAM_BA_FUSE_UL=0 COSLAM_BA_SEGTIME=1 timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop --only-solve joint > $O/joint_nofuse.json 2> $O/joint_nofuse.err
grep "ba segtime" $O/joint_nofuse.err | sed -n 30,38p
