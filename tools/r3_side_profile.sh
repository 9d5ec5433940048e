#!/bin/bash
# usage (GPU box): tools/r3_side_profile.sh  -> gpurun_out/r3_side_*.log : PPO step, rollout, bf16x3 step, MlpVAE step under rocprofv3 --kernel-trace --stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
python tools/ppo_probe.py > gpurun_out/r3_side_ppo.log 2>&1
bash tools/gpu_prof.sh "python $PWD/tools/ppo_probe.py" 12 >> gpurun_out/r3_side_ppo.log 2>&1
python tools/ppo_probe.py 2048 >> gpurun_out/r3_side_ppo.log 2>&1
python tools/rollout_latency.py > gpurun_out/r3_side_rollout.log 2>&1
bash tools/gpu_prof.sh "python $PWD/bench.py --precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --no-dp-form" 30 > gpurun_out/r3_side_x3.log 2>&1
python tools/mlp_vae_bench.py > gpurun_out/r3_side_mlp.log 2>&1
tail -3 gpurun_out/r3_side_ppo.log; tail -2 gpurun_out/r3_side_rollout.log; tail -2 gpurun_out/r3_side_mlp.log
