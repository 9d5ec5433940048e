#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
MI355_THIRD=1 MI355_PRECISION=bf16 timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -q -x -p no:cacheprovider -k "b512 or epoch or graph or adam" 2>&1 | tail -3
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2 3; do
for v in 0 1; do
  MI355_THIRD=$v timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); po=d['per_op_ms']; print('THIRD=$v', round(d['ms_per_step'],4), d['final_losses'])"
done; done
