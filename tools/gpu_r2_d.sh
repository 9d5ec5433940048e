#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "rwconv" > $O/pytest_ops.log 2>&1; tail -4 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "b512 or generations" > $O/pytest_vae.log 2>&1; tail -4 $O/pytest_vae.log
python tools/op_bench.py deconv3.fwd conv2.dgrad
for nb in 32 48 64 96; do echo "blocks/xcd $nb"; MI355_RWCONV_BLOCKS=$nb python tools/op_bench.py deconv3.fwd conv2.dgrad; done
X="--no-cpu-baseline --no-ppo --no-fp32 --no-replay"
MI355_RWCONV=1 timeout 300 python bench.py $X > $O/bench_rw1.json 2> $O/bench_rw1.err
for f in bench_rw1; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), "frames/s", round(d["ms_per_step"],4), "ms", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"]*1e3,1), "us", d["roofline"]["bound"], round(d["roofline"]["frac"],3))
    print({k:v for k,v in d["per_op_ms"].items() if k in ("deconv3.fwd","conv2.dgrad","conv2.fwd","deconv3.dgrad","deconv3.wgrad","deconv4.fwd")})
except Exception as e:
    print("$f", "ERR", e); print(open("$O/$f.err").read()[-600:])
PY
done
