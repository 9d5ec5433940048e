#!/bin/bash
# round 4, GPU call 1: whole -m gpu suite (no -x: every failure is wanted), trajectory tables, full bench line, MI355_KEVENT A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r4_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t1.log
timeout 300 python -m pytest tests/test_zz_adam_trajectory_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r4_traj.log 2>&1
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
for r in 1 2; do
for v in 0 1 2; do
  MI355_KEVENT=$v timeout 300 python bench.py $X 2>gpurun_out/r4_kev$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KEVENT=$v', round(d['ms_per_step'],4), round(d['value']))" >> gpurun_out/r4_ab1.log 2>&1
done; done
grep -v "^  File\|Extension modules" gpurun_out/r4_t1.log | tail -40
cat gpurun_out/r4_ab1.log
tail -c 1500 gpurun_out/r4_bench1.json
