#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=200 tools/ab_env.sh gpurun_out/r5_ab_ring2.txt 6 "MI355_DEFAULTS=1" "MI355_ENC12_RING=1" "MI355_LATENT_SPLIT=32"
