#!/bin/bash
# round-3 call: fused decoder tail -- op test first, then the whole GPU suite, then interleaved bench A/B (MI355_DECTAIL=0|1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "decoder_tail" > gpurun_out/pytest_r3c.log 2>&1; echo "op rc=$?" >> gpurun_out/pytest_r3c.log
tail -15 gpurun_out/pytest_r3c.log
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --steps 100 --warmup 10"
for r in 1 2; do for v in 0 1; do
  MI355_DECTAIL=$v timeout 200 python bench.py $X 2>gpurun_out/bench_dt$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); po=d['per_op_ms']; print('DECTAIL=$v', round(d['ms_per_step'],4), 'ms | ', ' '.join('%s %.1f'%(k,po[k]*1e3) for k in ('deconv4.fwd','deconv4.dgrad','deconv4.wgrad','deconv3.dgrad','deconv3.wgrad') if k in po), '| recon', d['final_losses'])"
done; done 2>&1 | tee gpurun_out/r3_ab2.log
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -rf -p no:cacheprovider > gpurun_out/pytest_r3d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r3d.log
tail -8 gpurun_out/pytest_r3d.log
