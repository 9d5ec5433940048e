#!/bin/bash
# usage: tools/gpu_seq.sh "<python command>" <kernels per repetition> [reps]  -> per-position durations and gaps of the trailing launch sequence
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_s -o p -- $1 > $R/gpurun_out/prof_s.log 2>&1
cd $R/tools
python rocpd_sequence.py $(find $R/gpurun_out/prof_s -name "*.db" | head -1) $2 ${3:-100} 2>&1
rm -rf $R/gpurun_out/prof_s
