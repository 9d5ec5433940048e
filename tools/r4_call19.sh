#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "whole_tiles" 2>&1 | tail -3
for r in 1 2 3; do
for n in 3 4; do
echo "NST=$n $(MI355_DWG_NST=$n timeout 300 python tools/mlp_vae_bench.py --steps 50 --precision bf16 2>&1 | tail -1)"
done; done
MI355_DWG_NST=3 bash tools/gpu_prof.sh "python $PWD/tools/mlp_vae_bench.py --steps 20 --precision bf16" 6 2>&1 | grep dwg
MI355_DWG_NST=4 bash tools/gpu_prof.sh "python $PWD/tools/mlp_vae_bench.py --steps 20 --precision bf16" 6 2>&1 | grep dwg
