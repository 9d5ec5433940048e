#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "colsum or bias or wgrad" 2>&1 | tail -3
for r in 1 2; do
  timeout 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); po=d['per_op_ms']; print('x3', round(d['ms_per_step'],4), round(d['value']), {k:round(v*1e3) for k,v in po.items() if 'bias' in k})"
done
timeout 300 python bench.py --precision fp32 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', round(d['ms_per_step'],4), round(d['value']))"
timeout 300 python bench.py --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', round(d['ms_per_step'],4), round(d['value']))"
