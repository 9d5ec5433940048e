#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3"
for r in 1 2 3; do
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 300" "--steps 200 --warmup 10"; do
ms=$(timeout 300 python bench.py $X $cfg 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],4))")
echo "$cfg | $ms"
done; done
