#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "wgrad" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -q -x -p no:cacheprovider -k "bf16x3" 2>&1 | tail -3
for r in 1 2; do
for v in 1 0; do
  MI355_X3_TAPWGRAD=$v timeout 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); po=d['per_op_ms']; print('X3TAP=$v', round(d['ms_per_step'],4), round(d['value']), ' '.join('%s %.0f'%(k,po.get(k,0)*1e3) for k in ('conv2.wgrad','conv3.wgrad','conv4.wgrad','deconv1.wgrad','deconv2.wgrad','deconv3.wgrad')), d['final_losses'])"
done; done
