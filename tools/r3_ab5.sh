#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "decoder_tail or full_batch or train_step_losses or captured_graph or kernel_generations or config1" 2>&1 | tail -5
timeout 200 python tools/dectail_bench.py 2>&1 | grep -v amdgpu.ids
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100 --warmup 10"
for r in 1 2 3; do
  timeout 200 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],4), 'ms', d['final_losses'])"
done 2>&1
