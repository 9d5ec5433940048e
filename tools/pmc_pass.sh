#!/bin/bash
# usage (GPU box): tools/pmc_pass.sh <tag> "<counters...>"  -> gpurun_out/pmc_<tag>.md  (one rocprofv3 --pmc pass of a short bench run)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc $@ -d $R/gpurun_out/pmc_$tag -o p -- python $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --no-dp-form --condition-ms 0 --no-box > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R
python tools/rocpd_pmc_summary.py $(find gpurun_out/pmc_$tag -name "*.db" | head -1) > gpurun_out/pmc_$tag.md 2>> gpurun_out/pmc_$tag.log
rm -rf gpurun_out/pmc_$tag
