#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/ab_env.sh gpurun_out/r4_ab18.log 3 "MI355_EVENT_SCOPE=0" "MI355_EVENT_SCOPE=1" "MI355_EVENT_SCOPE=2" > gpurun_out/r4_ab18.txt 2>&1
cat gpurun_out/r4_ab18.txt
MI355_EVENT_SCOPE=2 timeout 600 python -m pytest tests/test_a_c2_b512_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
