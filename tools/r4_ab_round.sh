#!/bin/bash
# one box, one library: every step-time change of round 4 switched off by its environment knob vs the defaults (three interleaved rounds, tools/ab_env.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/ab_env.sh gpurun_out/r4_ab_round.log 3 "MI355_KEVENT=0 MI355_ARES=0 MI355_ADAM_LAYOUTS=0 MI355_TAIL_FUSE=0 MI355_DENSE_BIAS_FUSED=0" "MI355_ARES=0" "MI355_KEVENT=0" "MI355_ARES=1" > gpurun_out/r4_ab_round.txt 2>&1
cat gpurun_out/r4_ab_round.txt
