#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100 --warmup 10"
run() { env "$@" timeout 200 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],4), 'ms')"; }
for r in 1 2 3; do
  run MI355_DECTAIL=0
  run MI355_DEFER=0 MI355_WGRAD_MAIN_MASK=5
  run MI355_DEFER=0 MI355_WGRAD_MAIN_MASK=1
  run MI355_DEFER=1 MI355_WGRAD_MAIN_MASK=5
  run MI355_DEFER=1 MI355_WGRAD_MAIN_MASK=1
done
