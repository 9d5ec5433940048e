#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "train_step_losses and bf16x3" 2>&1 | grep -E "^E  |assert|passed|failed" | head -12
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100 --warmup 10"
for r in 1 2; do for v in 0 1; do
  MI355_DECTAIL=$v timeout 200 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DECTAIL=$v step', round(d['ms_per_step'],4), 'ms')"
done; done 2>&1
