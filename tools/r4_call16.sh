#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_f_mlp_vae_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "whole_tiles or mlp or ordered_dense" 2>&1 | tail -12
timeout 300 python tools/mlp_vae_bench.py --steps 30 --precision bf16 2>&1 | tail -1
MI355_DWG=0 timeout 300 python tools/mlp_vae_bench.py --steps 30 --precision bf16 2>&1 | tail -1
bash tools/gpu_prof.sh "python $PWD/tools/mlp_vae_bench.py --steps 20 --precision bf16" 14 > gpurun_out/r4_mlp_prof3.log 2>&1
cat gpurun_out/r4_mlp_prof3.log
