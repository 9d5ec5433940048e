#!/bin/bash
# round 4, GPU call 4: pair-unit tiled reduce, dense filter gradients without row splits, graph path removed: tests, A/B, timeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_a_c2_b512_gpu.py tests/test_ops_gpu.py tests/test_vae_gpu.py tests/test_ref_graph_gpu.py tests/test_zz_adam_trajectory_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/r4_t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t4.log
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
rm -f gpurun_out/r4_ab4.log
for r in 1 2; do
for v in "256 1" "96 1" "96 0" "256 0"; do
  set -- $v
  MI355_DENSE_WGRAD_BLOCKS=$1 MI355_TAIL_FUSE=$2 timeout 300 python bench.py $X 2>gpurun_out/r4_ab4.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DENSE_WGRAD_BLOCKS=$1 TAIL_FUSE=$2', round(d['ms_per_step'],4), round(d['value']))" >> gpurun_out/r4_ab4.log 2>&1
done; done
MI355_DENSE_WGRAD_BLOCKS=96 tools/timeline.sh r04c
grep -v "^  File\|Extension modules" gpurun_out/r4_t4.log | tail -12
cat gpurun_out/r4_ab4.log
sed -n 28,45p gpurun_out/timeline_r04c.md
