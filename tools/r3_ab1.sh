#!/bin/bash
# round-3 A/B call 1: small-grid tile variants, bf16x3 stream placement, re-run of the bf16x3 parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== gemm2 tile (key 17: 2 = 128x64, 1 = 64x64)"
for v in 2 1 2 1; do timeout 120 python tools/op_bench.py conv4.fwd deconv1.dgrad --set 17=$v --iters 100 | sed "s/^/17=$v  /"; done
echo "== tapconv tile (key 5: 0 auto, 2 = 128-position tile)"
for v in 0 2 0 2; do timeout 120 python tools/op_bench.py deconv1.fwd conv4.dgrad --set 5=$v --iters 100 | sed "s/^/5=$v  /"; done
echo "== bf16 step under key 17"
timeout 200 python tools/ab_step.py 17 2 1
echo "== bf16 step under key 5"
timeout 200 python tools/ab_step.py 5 0 2
} > gpurun_out/r3_ab1.log 2>&1
X="--precision bf16x3 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 30 --warmup 5"
for mm in 5 0 15 7; do
  MI355_WGRAD_MAIN_MASK=$mm timeout 200 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3 main_mask=$mm', round(d['ms_per_step'],4))" >> gpurun_out/r3_ab1.log 2>&1
done
MI355_BWD_STREAMS=0 timeout 200 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3 one stream', round(d['ms_per_step'],4))" >> gpurun_out/r3_ab1.log 2>&1
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "bf16x3 or x3 or split or config3" > gpurun_out/pytest_r3b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r3b.log
tail -3 gpurun_out/pytest_r3b.log
cat gpurun_out/r3_ab1.log
