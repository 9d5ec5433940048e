#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_d_c4_dp_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "dense or ordered or comm or parallel" > gpurun_out/r4_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t5.log
tools/ab_env.sh gpurun_out/r4_ab5.log 3 "MI355_DENSE_WGRAD_BLOCKS=256" "MI355_DENSE_WGRAD_BLOCKS=512" "MI355_DENSE_WGRAD_BLOCKS=1024" "MI355_DENSE_WGRAD_BLOCKS=1024 MI355_TAIL_FUSE=0" "MI355_DENSE_WGRAD_BLOCKS=1024 MI355_KEVENT=0" > gpurun_out/r4_ab5.txt 2>&1
MI355_DENSE_WGRAD_BLOCKS=1024 tools/timeline.sh r04d
tail -5 gpurun_out/r4_t5.log
cat gpurun_out/r4_ab5.txt
sed -n 28,45p gpurun_out/timeline_r04d.md
