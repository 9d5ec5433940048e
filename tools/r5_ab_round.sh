#!/bin/bash
# usage (GPU box, one gpurun call): tools/r5_ab_round.sh  -> gpurun_out/r5_ab_round.txt : every step-time change of round 5 off / on, on ONE box and ONE library
# (200 steps each, four interleaved rounds, medians): the fused encoder head of the forward pass (enc12), the decoder slab sums in the filter-gradient queue's idle gap,
# finalize_losses behind that queue's last reduce, the fragment-ordered weight copies out of the Adam launch -- and each of them alone
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OFF="MI355_ENC12=0 MI355_MID_FLUSH=0 MI355_FIN_SIDE=0 MI355_ADAM_FRAG=0"
tools/ab_env.sh gpurun_out/r5_ab_round.txt 4 "$OFF" "MI355_ENC12=1 MI355_MID_FLUSH=0 MI355_FIN_SIDE=0 MI355_ADAM_FRAG=0" "MI355_ENC12=0 MI355_MID_FLUSH=1 MI355_FIN_SIDE=1 MI355_ADAM_FRAG=0" "MI355_ENC12=0 MI355_MID_FLUSH=0 MI355_FIN_SIDE=0 MI355_ADAM_FRAG=1" "MI355_DEFAULTS=1"
