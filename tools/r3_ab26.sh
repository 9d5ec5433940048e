#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2; do
for v in "1 1 1" "0 1 1" "1 0 1" "0 0 0"; do
  set -- $v
  MI355_DECTAIL=$1 MI355_ENCHEAD=$2 MI355_SLAB_BF16=$3 timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DECTAIL=$1 ENCHEAD=$2 SLAB16=$3', round(d['ms_per_step'],4), round(d['value']))"
done; done
