#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r05k}
timeout 900 python -m pytest tests/test_f_mlp_vae_gpu.py tests/test_ops_gpu.py -k "mlp or gather_rows or recon_loss or reparam" -x -q -p no:cacheprovider > gpurun_out/new_tests_$tag.log 2>&1
echo "new tests rc=$?"; tail -6 gpurun_out/new_tests_$tag.log
for r in 1 2 3; do for f in f32 u8; do echo "$(timeout 200 python tools/mlp_vae_bench.py --steps 100 --precision bf16 --frames $f 2>/dev/null | tail -1)" | tee -a gpurun_out/mlp_u8_$tag.txt; done; done
