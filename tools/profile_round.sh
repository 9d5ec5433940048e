#!/bin/bash
# usage (GPU box, one gpurun call): tools/profile_round.sh <tag>
#   -> gpurun_out/{bench_<tag>_full.json, prof_<tag>.md, pmc_<tag>_{sq,fetch,write}.md, timeline_<tag>.md}; then (anywhere):
#      python tools/make_profile.py <tag> <out-name> "<title>"
tag=$1
R=$GRAFT_REPO_ROOT
cd $R
timeout 400 python bench.py --steps 30 --warmup 5 > gpurun_out/bench_${tag}_full.json 2> gpurun_out/bench_${tag}_full.err
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --no-dp-form --no-box > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/prof_$tag -name "*.db" | head -1) > gpurun_out/prof_$tag.md 2>> gpurun_out/prof_$tag.log
rm -rf gpurun_out/prof_$tag
tools/pmc_pass.sh ${tag}_sq GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES
tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE
tools/pmc_pass.sh ${tag}_write WRITE_SIZE
tools/timeline.sh $tag
tail -c 600 gpurun_out/bench_${tag}_full.json
