#!/bin/bash
# round 4, GPU call 2: suite after the fused tail reduce, (KEVENT x TAIL_FUSE) A/B, dispatch timeline + kernel stats of the default configuration
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/r4_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t2.log
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100"
rm -f gpurun_out/r4_ab2.log
for r in 1 2; do
for v in "0 0" "1 0" "0 1" "1 1"; do
  set -- $v
  MI355_KEVENT=$1 MI355_TAIL_FUSE=$2 timeout 300 python bench.py $X 2>gpurun_out/r4_ab2.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KEVENT=$1 TAIL_FUSE=$2', round(d['ms_per_step'],4), round(d['value']))" >> gpurun_out/r4_ab2.log 2>&1
done; done
tools/timeline.sh r04a
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04a -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 > $R/gpurun_out/prof_r04a.log 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/prof_r04a -name "*.db" | head -1) > gpurun_out/prof_r04a.md 2>> gpurun_out/prof_r04a.log
rm -rf gpurun_out/prof_r04a
grep -v "^  File\|Extension modules" gpurun_out/r4_t2.log | tail -15
cat gpurun_out/r4_ab2.log
