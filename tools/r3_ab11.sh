#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "decoder_tail or (wgrad and bf16) or full_batch or train_step_losses or captured_graph or kernel_generations" 2>&1 | tail -3
timeout 200 python tools/dectail_bench.py 2>&1 | grep -v amdgpu.ids
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100 --warmup 10"
run() { env "$@" timeout 200 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],4), 'ms', d['final_losses'])"; }
for r in 1 2 3; do
  run MI355_DECTAIL=0
  run MI355_SLAB_BF16=0
  run MI355_SLAB_BF16=1
done
