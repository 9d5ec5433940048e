#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_a_c2_b512_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
MI355_DENSE_EARLY=1 timeout 600 python -m pytest tests/test_a_c2_b512_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "bitwise or b512_train" 2>&1 | tail -3
tools/ab_env.sh gpurun_out/r4_ab9.log 3 "MI355_DENSE_EARLY=0" "MI355_DENSE_EARLY=1" "MI355_KEVENT=3" "MI355_KEVENT=3 MI355_DENSE_EARLY=1" "MI355_KEVENT=0 MI355_DENSE_EARLY=1" > gpurun_out/r4_ab9.txt 2>&1
cat gpurun_out/r4_ab9.txt
MI355_DENSE_EARLY=1 tools/timeline.sh r04g
sed -n 26,42p gpurun_out/timeline_r04g.md
