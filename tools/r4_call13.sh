#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "activation_resident" 2>&1 | tail -12
timeout 300 python tools/ares_bench.py 512 2>&1 | tail -4
timeout 900 python -m pytest tests/test_a_c2_b512_gpu.py tests/test_ref_graph_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
tools/ab_env.sh gpurun_out/r4_ab13.log 3 "MI355_ARES_MID=0" "MI355_ARES_MID=1" > gpurun_out/r4_ab13.txt 2>&1
cat gpurun_out/r4_ab13.txt
