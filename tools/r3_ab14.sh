#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "decoder_tail" 2>&1 | tail -5
timeout 120 python tools/dectail_bench.py --iters 50 2>&1 | tail -6
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2; do
for v in 1 0; do
  MI355_DECTAIL=$v timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DECTAIL=$v', round(d['ms_per_step'],4), 'tail', d['per_op_ms'].get('deconv4.fwd'), d['final_losses'])"
done; done
