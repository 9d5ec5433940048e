#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
MI355_ARES_CFG=5 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "activation_resident" 2>&1 | tail -3
MI355_ARES_CFG=1 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "activation_resident" 2>&1 | tail -2
for cfg in 1 5; do echo "cfg $cfg"; MI355_ARES_CFG=$cfg timeout 300 python tools/ares_bench.py 512 2>&1 | tail -2; done
tools/ab_env.sh gpurun_out/r4_ab10.log 3 "MI355_ARES=0" "MI355_ARES_CFG=1" "MI355_ARES_CFG=5" > gpurun_out/r4_ab10.txt 2>&1
cat gpurun_out/r4_ab10.txt
