#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "fused_minibatch or replay or config3 or fused_step" 2>&1 | tail -4
for v in 0 1; do
MI355_PPO_IDX=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 --no-mlp --no-x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('IDX=$v', d['ppo']); r=d['replay']; print({k:r[k] for k in ('seconds','sgd_s','ppo_sgd_samples_per_s','last_loss')})"
done
