#!/bin/bash
# one gpurun call: the real-RCCL one-rank DP tests + the latent-layer tests on the rebuilt library, the MlpVAE profile passes, then the A/B of the knobs added late in round 5
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_d_c4_dp_gpu.py -q -x -p no:cacheprovider -k "real_rccl or one_c_call or one_rank" > gpurun_out/call_m_tests.log 2>&1; echo "dp tests rc=$?"; tail -3 gpurun_out/call_m_tests.log
MI355_REPARAM_WIDE=1 MI355_LATENT_SPLIT=16 timeout 600 python -m pytest tests/test_vae_gpu.py -q -x -p no:cacheprovider > gpurun_out/call_m_tests_knobs.log 2>&1; echo "vae tests under knobs rc=$?"; tail -3 gpurun_out/call_m_tests_knobs.log
tools/mlp_profile.sh r05b
STEPS=200 tools/ab_env.sh gpurun_out/r5_ab_late.txt 3 "MI355_DEFAULTS=1" "MI355_SIDE_PRIO=-1" "MI355_SIDE_PRIO=1" "MI355_REPARAM_WIDE=1" "MI355_LATENT_SPLIT=16" "MI355_LATENT_SPLIT=8" "MI355_REPARAM_WIDE=1 MI355_LATENT_SPLIT=16"
