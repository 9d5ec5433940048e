#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "activation_resident" 2>&1 | tail -3
timeout 300 python tools/ares_bench.py 512 2>&1 | tail -3 | tee gpurun_out/r4_ares_bench.txt
tools/ab_env.sh gpurun_out/r4_ab7.log 3 "MI355_ARES=0" "MI355_ARES=1" > gpurun_out/r4_ab7.txt 2>&1
cat gpurun_out/r4_ab7.txt
tools/timeline.sh r04f
sed -n 1,20p gpurun_out/timeline_r04f.md
