#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_f_mlp_vae_gpu.py tests/test_a_c2_b512_gpu.py tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "mlp or adam or gather_rows or ordered_dense or dense_gemm or b512 or bitwise" 2>&1 | tail -8
timeout 300 python tools/mlp_vae_bench.py --steps 30 2>&1 | tail -2
MI355_GEMM2_SPLITK=0 timeout 300 python tools/mlp_vae_bench.py --steps 30 --precision bf16 2>&1 | tail -1
bash tools/gpu_prof.sh "python $PWD/tools/mlp_vae_bench.py --steps 20 --precision bf16" 16 > gpurun_out/r4_mlp_prof2.log 2>&1
cat gpurun_out/r4_mlp_prof2.log
