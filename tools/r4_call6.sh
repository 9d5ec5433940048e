#!/bin/bash
# round 4, GPU call 6: activation-resident kernels for the four small-grid layers: op test, model tests, A/B MI355_ARES, timeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "activation_resident" > gpurun_out/r4_t6a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t6a.log
tail -30 gpurun_out/r4_t6a.log
timeout 900 python -m pytest tests/test_a_c2_b512_gpu.py tests/test_b_c1_epoch_gpu.py tests/test_ref_graph_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/r4_t6b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t6b.log
tail -12 gpurun_out/r4_t6b.log
tools/ab_env.sh gpurun_out/r4_ab6.log 3 "MI355_ARES=0" "MI355_ARES=1" > gpurun_out/r4_ab6.txt 2>&1
cat gpurun_out/r4_ab6.txt
tools/timeline.sh r04e
cat gpurun_out/timeline_r04e.md | head -45
