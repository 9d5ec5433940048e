cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for r in 1 2; do
for cfg in "A=0" "MI355_GEMM2_STAGES=3" "MI355_GEMM2_TILE=3 MI355_GEMM2_STAGES=3" "MI355_GEMM2_TILE=3" "MI355_DWG_NST=4" "MI355_GEMM2_SPLITK=0" "MI355_MLP_STREAMS=0"; do
  echo "$cfg | $(env $cfg timeout 200 python tools/mlp_vae_bench.py --steps 100 --precision bf16 2>/dev/null | tail -1)" | tee -a gpurun_out/mlp_knobs_r05f.txt
done; done
