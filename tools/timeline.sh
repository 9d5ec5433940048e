#!/bin/bash
# usage (GPU box): tools/timeline.sh <tag>  -> gpurun_out/timeline_<tag>.md : one training step's dispatches (rocprofv3 --kernel-trace)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$tag -o t -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --no-dp-form --condition-ms 0 --no-box > $R/gpurun_out/tl_$tag.log 2>&1
cd $R
python tools/rocpd_timeline.py $(find gpurun_out/tl_$tag -name "*.db" | head -1) > gpurun_out/timeline_$tag.md
rm -rf gpurun_out/tl_$tag
