#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "decoder_tail" 2>&1 | tail -4
timeout 200 python tools/dectail_bench.py 2>&1 | grep -v amdgpu.ids
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100 --warmup 10"
for r in 1 2; do for v in 0 1; do
  MI355_DECTAIL=$v timeout 200 python bench.py $X 2>gpurun_out/bench_dt$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); po=d['per_op_ms']; print('DECTAIL=$v', round(d['ms_per_step'],4), 'ms | ', ' '.join('%s %.1f'%(k,po[k]*1e3) for k in ('deconv4.fwd','deconv4.dgrad','deconv4.wgrad','deconv3.dgrad','deconv3.wgrad') if k in po))"
done; done 2>&1
