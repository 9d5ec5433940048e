#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_f_mlp_vae_gpu.py tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "mlp or adam or gather_rows or ordered_dense" 2>&1 | tail -15
timeout 300 python tools/mlp_vae_bench.py --steps 30 2>&1 | tail -3
bash tools/gpu_prof.sh "python $PWD/tools/mlp_vae_bench.py --steps 20 --precision bf16" 30 > gpurun_out/r4_mlp_prof.log 2>&1
cat gpurun_out/r4_mlp_prof.log
