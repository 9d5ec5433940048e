#!/bin/bash
# one gpurun call: sweep of existing knobs under the end-of-round-5 schedule (three interleaved rounds, 200 steps each)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=200 tools/ab_env.sh gpurun_out/r5_ab_sweep.txt 3 "MI355_DEFAULTS=1" "MI355_LATENT_SPLIT=12" "MI355_LATENT_SPLIT=16" "MI355_LATENT_SPLIT=24" \
  "MI355_DENSE_WGRAD_BLOCKS=96" "MI355_DENSE_WGRAD_BLOCKS=128" "MI355_DENSE_WGRAD_BLOCKS=192" "MI355_DENSE_WGRAD_BLOCKS=384" \
  "MI355_EVENT_SCOPE=1" "MI355_EVENT_SCOPE=2" "MI355_HEADS_MAIN=0" "MI355_THIRD=1" "MI355_ARES_CFG=1" "MI355_ARES_CFG=2" "MI355_LATENT_SPLIT=16 MI355_DENSE_WGRAD_BLOCKS=128"
