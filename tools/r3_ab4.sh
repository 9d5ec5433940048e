#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100 --warmup 10"
for r in 1 2; do for mm in 5 1 4 0 3; do
  MI355_WGRAD_MAIN_MASK=$mm timeout 200 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('main_mask=$mm', round(d['ms_per_step'],4), 'ms')"
done; done 2>&1
