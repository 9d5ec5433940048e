#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r05j}
timeout 900 python -m pytest tests/test_ops_gpu.py -k "dense_wgrad or ordered_dense" tests/test_a_c2_b512_gpu.py tests/test_f_mlp_vae_gpu.py -x -q -p no:cacheprovider > gpurun_out/new_tests_$tag.log 2>&1
echo "new tests rc=$?"; tail -4 gpurun_out/new_tests_$tag.log
tools/ab_env.sh gpurun_out/ab_$tag.txt 4 "MI355_FIN_SIDE=0" "MI355_FIN_SIDE=1"
tools/timeline.sh $tag; sed -n 36,48p gpurun_out/timeline_$tag.md
