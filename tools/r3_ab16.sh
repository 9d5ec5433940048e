#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for s in 3 4; do
MI355_GEMM2_STAGES=$s timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv_fwd or deconv_fwd or dense or generations" 2>&1 | tail -2
done
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2 3; do
for v in 2 3 4; do
  MI355_GEMM2_STAGES=$v timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); po=d['per_op_ms']; print('STAGES=$v', round(d['ms_per_step'],4), ' '.join('%s %.1f'%(k,po[k]*1e3) for k in ('conv4.fwd','deconv1.dgrad','dense1.fwd','heads.dgrad','dense1.dgrad')), d['final_losses'])"
done; done
