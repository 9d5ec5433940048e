#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_f_mlp_vae_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
for r in 1 2 3; do for v in 0 1; do
echo "STREAMS=$v $(MI355_MLP_STREAMS=$v timeout 300 python tools/mlp_vae_bench.py --steps 50 --precision bf16 2>&1 | tail -1)"
done; done
MI355_MLP_STREAMS=1 timeout 300 python tools/mlp_vae_bench.py --steps 30 --precision fp32 2>&1 | tail -1
