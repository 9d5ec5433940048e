#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r05l}
timeout 900 python -m pytest tests/test_ops_gpu.py -k "encoder_head_forward" tests/test_f_mlp_vae_gpu.py -x -q -p no:cacheprovider > gpurun_out/new_tests_$tag.log 2>&1
echo "new tests rc=$?"; tail -8 gpurun_out/new_tests_$tag.log
tools/ab_env.sh gpurun_out/ab_$tag.txt 3 "MI355_ENC12=0" "MI355_ENC12=1"
tools/timeline.sh $tag; sed -n 1,12p gpurun_out/timeline_$tag.md
timeout 900 python -m pytest tests/test_a_c2_b512_gpu.py tests/test_b_c1_epoch_gpu.py tests/test_vae_gpu.py -x -q -p no:cacheprovider > gpurun_out/model_tests_$tag.log 2>&1
echo "model tests rc=$?"; tail -4 gpurun_out/model_tests_$tag.log
