#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r05i}
timeout 900 python -m pytest tests/test_ops_gpu.py -k "dense_wgrad or ordered_dense" tests/test_f_mlp_vae_gpu.py -x -q -p no:cacheprovider > gpurun_out/new_tests_$tag.log 2>&1
echo "new tests rc=$?"; tail -5 gpurun_out/new_tests_$tag.log
tools/ab_env.sh gpurun_out/ab_$tag.txt 3 "MI355_DWGS=0" "MI355_DWGS=1"
for r in 1 2 3; do for v in 0 1; do echo "DWGS=$v $(MI355_DWGS=$v timeout 200 python tools/mlp_vae_bench.py --steps 100 --precision bf16 2>/dev/null | tail -1)" | tee -a gpurun_out/mlp_dwgs_$tag.txt; done; done
tools/timeline.sh $tag; sed -n 38,48p gpurun_out/timeline_$tag.md
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/suite_$tag.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/suite_$tag.log | tail -3
