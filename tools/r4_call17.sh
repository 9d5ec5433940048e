#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_a_c2_b512_gpu.py tests/test_b_c1_epoch_gpu.py tests/test_ops_gpu.py tests/test_zz_adam_trajectory_gpu.py tests/test_ref_graph_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "b512 or bitwise or epoch or partial_sum_slabs or trajectory or adam or ref_graph or matches" 2>&1 | tail -8
tools/ab_env.sh gpurun_out/r4_ab17.log 3 "MI355_ADAM_LAYOUTS=0" "MI355_ADAM_LAYOUTS=1" > gpurun_out/r4_ab17.txt 2>&1
cat gpurun_out/r4_ab17.txt
