#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "decoder_tail" 2>&1 | tail -3
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2; do
for v in "0 1" "1 1" "1 0" "0 0"; do
  set -- $v
  export MI355_DECTAIL_ROTATE=$1 MI355_DECTAIL_EDGE=$2
  timeout 120 python tools/dectail_bench.py --iters 50 2>&1 | grep -i "fused" | head -2
  timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ROT=$1 EDGE=$2', round(d['ms_per_step'],4), 'tail', d['per_op_ms'].get('deconv4.fwd'))"
done; done
