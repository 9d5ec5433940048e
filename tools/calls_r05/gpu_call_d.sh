#!/bin/bash
# usage (GPU box): tools/gpu_call_d.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r05d}
timeout 900 python -m pytest tests/test_ops_gpu.py -k "wgrad or ordered_dense or bf16_partial" tests/test_h_script_traces_gpu.py -x -q -p no:cacheprovider > gpurun_out/new_tests_$tag.log 2>&1
echo "new tests rc=$?"; tail -12 gpurun_out/new_tests_$tag.log
tools/ab_env.sh gpurun_out/ab_$tag.txt 3 "MI355_DWGS=0 MI355_SLAB_TR=0" "MI355_DWGS=0 MI355_SLAB_TR=1" "MI355_DWGS=1 MI355_SLAB_TR=1" "MI355_DWGS=1 MI355_SLAB_TR=0"
tools/timeline.sh $tag; sed -n 28,42p gpurun_out/timeline_$tag.md
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/suite_$tag.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/suite_$tag.log | tail -3
