#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r05h}
tools/ab_env.sh gpurun_out/ab_$tag.txt 4 "MI355_MID_FLUSH=0" "MI355_MID_FLUSH=1"
tools/timeline.sh $tag; sed -n 14,42p gpurun_out/timeline_$tag.md
