#!/bin/bash
# usage (GPU box, one gpurun call): tools/r5_ab_round.sh  -> gpurun_out/r5_ab_round.txt : every step-time change of round 5 off / on, on ONE box and ONE library
# (200 steps each, four interleaved rounds, medians): the fused encoder head of the forward pass (enc12; then its hand-scheduled loads and LDS reads), the decoder slab sums in
# the filter-gradient queue's idle gap, finalize_losses behind that queue's last reduce, the fragment-ordered weight copies out of the Adam launch, the latent layers' split
# cap -- and each group alone
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
E12="MI355_ENC12_RING=0 MI355_ENC12_C2=0"
OFF="MI355_ENC12=0 MI355_MID_FLUSH=0 MI355_FIN_SIDE=0 MI355_ADAM_FRAG=0 MI355_LATENT_SPLIT=32"
tools/ab_env.sh gpurun_out/r5_ab_round.txt 4 "$OFF" "MI355_ENC12=1 $E12 MI355_MID_FLUSH=0 MI355_FIN_SIDE=0 MI355_ADAM_FRAG=0 MI355_LATENT_SPLIT=32" \
  "MI355_ENC12=1 MI355_MID_FLUSH=0 MI355_FIN_SIDE=0 MI355_ADAM_FRAG=0 MI355_LATENT_SPLIT=32" \
  "MI355_ENC12=0 MI355_MID_FLUSH=1 MI355_FIN_SIDE=1 MI355_ADAM_FRAG=0 MI355_LATENT_SPLIT=32" "MI355_ENC12=0 MI355_MID_FLUSH=0 MI355_FIN_SIDE=0 MI355_ADAM_FRAG=1 MI355_LATENT_SPLIT=32" \
  "MI355_ENC12=0 MI355_MID_FLUSH=0 MI355_FIN_SIDE=0 MI355_ADAM_FRAG=0 MI355_LATENT_SPLIT=16" "MI355_DEFAULTS=1"
