#!/bin/bash
# usage (GPU box): tools/gpu_call_e.sh <tag>   A/B runs only (no suite): placement of the latent layers' filter gradients x kernel, MlpVAE with / without dwgs, bf16x3 filter gradients
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r05e}
tools/ab_env.sh gpurun_out/ab_$tag.txt 3 "MI355_DWGS=0 MI355_LATE_DENSE=1" "MI355_DWGS=1 MI355_LATE_DENSE=1" "MI355_DWGS=0 MI355_LATE_DENSE=0" "MI355_DWGS=1 MI355_LATE_DENSE=0"
for r in 1 2 3; do for v in 0 1; do echo "DWGS=$v $(MI355_DWGS=$v timeout 200 python tools/mlp_vae_bench.py --steps 100 --precision bf16 2>/dev/null | tail -1)" | tee -a gpurun_out/mlp_dwgs_$tag.txt; done; done
STEPS=60 EXTRA="--precision bf16x3" tools/ab_env.sh gpurun_out/ab_x3_$tag.txt 2 "MI355_X3_TAPWGRAD=0" "MI355_X3_TAPWGRAD=1" "MI355_X3_TAPWGRAD=2"
