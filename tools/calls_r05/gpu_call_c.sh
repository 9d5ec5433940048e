#!/bin/bash
# usage (GPU box): tools/gpu_call_c.sh <tag>  -- dense filter-gradient tests + the round's new tests, A/B of the LDS-free dense filter gradient (MI355_DWGS) on the step,
# the PPO kernel-boundary measurement (MI355_PPO_PAD), the step timeline, the whole suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r05c}
timeout 900 python -m pytest tests/test_ops_gpu.py -k "dense_wgrad or ordered_dense" tests/test_g_box_probe_gpu.py tests/test_h_script_traces_gpu.py tests/test_f_mlp_vae_gpu.py -x -q -p no:cacheprovider > gpurun_out/new_tests_$tag.log 2>&1
echo "new tests rc=$?"; tail -12 gpurun_out/new_tests_$tag.log
tools/ab_env.sh gpurun_out/dwgs_ab_$tag.txt 3 "MI355_DWGS=0" "MI355_DWGS=1" "MI355_DWGS=1 MI355_DENSE_EARLY=1"
for r in 1 2 3; do for n in 0 2 4; do echo "PAD=$n $(MI355_PPO_PAD=$n timeout 100 python tools/ppo_probe.py 32 2>/dev/null | tail -1)" | tee -a gpurun_out/ppo_pad_$tag.txt; done; done
tools/timeline.sh $tag; sed -n 1,50p gpurun_out/timeline_$tag.md
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/suite_$tag.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/suite_$tag.log | tail -3
