#!/bin/bash
# usage: tools/gpu_prof.sh "<python command>"  -> per-kernel stats of that command (rocprofv3 --kernel-trace --stats)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o p -- $1 > $R/gpurun_out/prof_x.log 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/prof_x -name "*.db" | head -1) 2>&1 | head -${2:-25}
rm -rf gpurun_out/prof_x
