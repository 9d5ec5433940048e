#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for r in 1 2; do
for v in 0 1 2; do
  MI355_X3_TAPWGRAD=$v timeout 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('X3TAP=$v', round(d['ms_per_step'],4), round(d['value']))"
done; done
