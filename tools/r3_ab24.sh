#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2 3; do
for v in "1 0" "0 0" "1 1"; do
  set -- $v
  MI355_HEADS_MAIN=$1 MI355_THIRD=$2 timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HEADS_MAIN=$1 THIRD=$2', round(d['ms_per_step'],4))"
done; done
