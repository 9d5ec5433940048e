#!/bin/bash
# round 4: whole -m gpu suite as the driver runs it, full bench line, kernel stats + timeline of the current default configuration
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider > gpurun_out/r4_t12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t12.log
grep -v "^  File\|Extension modules" gpurun_out/r4_t12.log | tail -8
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/r4_bench12.json 2> gpurun_out/r4_bench12.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench12.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
print({k:(v.get('grad_worst'), v.get('adam_off_fraction')) for k,v in d['parity'].items() if isinstance(v, dict)})
print(d['bf16x3']['frames_per_s'], d['fp32']['frames_per_s'], d['mlp_vae']['ms_per_step'], d['ppo']['ms_per_update'], d['replay']['resident']['seconds'], d['replay']['from_host']['seconds'])
print(d['per_op_ms'])
PY
tools/timeline.sh r04h
R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04h -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 > $R/gpurun_out/prof_r04h.log 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/prof_r04h -name "*.db" | head -1) > gpurun_out/prof_r04h.md 2>> gpurun_out/prof_r04h.log
rm -rf gpurun_out/prof_r04h
head -40 gpurun_out/prof_r04h.md
