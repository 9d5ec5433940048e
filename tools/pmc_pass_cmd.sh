#!/bin/bash
# usage (GPU box): tools/pmc_pass_cmd.sh <tag> "<python command>" <counters...>  -> gpurun_out/pmc_<tag>.md  (one rocprofv3 --pmc pass of any command; no trace domains)
tag=$1; cmd=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc $@ -d $R/gpurun_out/pmc_$tag -o p -- $cmd > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R
python tools/rocpd_pmc_summary.py $(find gpurun_out/pmc_$tag -name "*.db" | head -1) > gpurun_out/pmc_$tag.md 2>> gpurun_out/pmc_$tag.log
rm -rf gpurun_out/pmc_$tag
