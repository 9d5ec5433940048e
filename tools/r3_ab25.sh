#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_vae_gpu.py -m gpu -q -x -p no:cacheprovider -k "uint8 or narrow or conv_fwd or relu_bit or b512 or encoder_head" 2>&1 | tail -3
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2 3; do
  timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); po=d['per_op_ms']; print(round(d['ms_per_step'],4), 'conv1.fwd', po['conv1.fwd'], d['final_losses'])"
done
