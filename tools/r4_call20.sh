#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_f_mlp_vae_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "adam or mlp" 2>&1 | tail -4
for r in 1 2; do
echo "$(timeout 300 python tools/mlp_vae_bench.py --steps 50 --precision bf16 2>&1 | tail -1)"
done
