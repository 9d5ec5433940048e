#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
MI355_NW_DEPTH=6 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_vae_gpu.py -m gpu -q -x -p no:cacheprovider -k "uint8 or narrow or b512" 2>&1 | tail -3
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2 3; do
for v in 3 6; do
  MI355_NW_DEPTH=$v timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEPTH=$v', round(d['ms_per_step'],4), 'conv1.wgrad', d['per_op_ms'].get('conv1.wgrad'), 'tail', d['per_op_ms'].get('deconv4.fwd'))"
done; done
