#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "rwconv or bit_words" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "b512 or generations or train_step_losses" 2>&1 | tail -8
python tools/op_bench.py deconv3.fwd conv2.dgrad
X="--no-cpu-baseline --no-ppo --no-fp32 --no-replay"
timeout 300 python bench.py $X > gpurun_out/bench_rw.json 2> gpurun_out/bench_rw.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_rw.json").read().strip().splitlines()[-1])
    print(round(d["value"]), "frames/s", round(d["ms_per_step"],4), "ms", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"]*1e3,1), "us", d["roofline"]["bound"], round(d["roofline"]["frac"],3))
    print(d["per_op_ms"])
except Exception as e:
    print("ERR", e); print(open("gpurun_out/bench_rw.err").read()[-800:])
PY
