#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2; do
for v in 1 2 4 8 6 12; do
  MI355_WGRAD_MAIN_MASK=$v timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); po=d['per_op_ms']; print('MASK=$v', round(d['ms_per_step'],4), ' '.join('%s %.1f'%(k,po.get(k,0)*1e3) for k in ('conv2.dgrad','conv2.wgrad','conv3.wgrad','conv4.wgrad')))"
done; done
