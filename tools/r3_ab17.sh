#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
MI355_GEMM2_TILE=3 MI355_GEMM2_STAGES=3 timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv_fwd or deconv_fwd or dense or generations" 2>&1 | tail -2
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2; do
for v in "2 2" "3 2" "3 3"; do
  set -- $v
  MI355_GEMM2_TILE=$1 MI355_GEMM2_STAGES=$2 timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); po=d['per_op_ms']; print('TILE=$1 STAGES=$2', round(d['ms_per_step'],4), ' '.join('%s %.1f'%(k,po[k]*1e3) for k in ('conv4.fwd','deconv1.dgrad','dense1.fwd','heads.dgrad','dense1.dgrad')))"
done; done
