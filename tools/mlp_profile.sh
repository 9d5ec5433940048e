#!/bin/bash
# usage (GPU box, one gpurun call): tools/mlp_profile.sh <tag>  -> gpurun_out/{mlp_prof_<tag>.md, pmc_mlp_<tag>_{sq,fetch,write}.md, mlp_bench_<tag>.txt}
# the MlpVAE SGD step (tools/mlp_vae_bench.py, bf16, batch 512, uint8 frame table): kernel times by rocprofv3 --kernel-trace --stats, then three separate --pmc passes
# (SQ / GRBM counters; FETCH_SIZE alone; WRITE_SIZE alone -- no trace domain combined with --pmc)
tag=$1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
CMD="python $R/tools/mlp_vae_bench.py --precision bf16 --steps 20"
timeout 200 $CMD > gpurun_out/mlp_bench_$tag.txt 2>&1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mlp_prof_$tag -o p -- $CMD > $R/gpurun_out/mlp_prof_$tag.log 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/mlp_prof_$tag -name "*.db" | head -1) > gpurun_out/mlp_prof_$tag.md 2>> gpurun_out/mlp_prof_$tag.log
rm -rf gpurun_out/mlp_prof_$tag
CMD3="python $R/tools/mlp_vae_bench.py --precision bf16 --steps 3"
tools/pmc_pass_cmd.sh mlp_${tag}_sq "$CMD3" GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES
tools/pmc_pass_cmd.sh mlp_${tag}_fetch "$CMD3" FETCH_SIZE
tools/pmc_pass_cmd.sh mlp_${tag}_write "$CMD3" WRITE_SIZE
cat gpurun_out/mlp_bench_$tag.txt | tail -2
head -20 gpurun_out/mlp_prof_$tag.md
