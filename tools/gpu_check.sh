#!/bin/bash
# usage (on the GPU box via gpurun): tools/gpu_check.sh <tag> [pytest args...]  -> tests summary + bench summary
tag=$1; shift
python -m pytest tests/test_ops_gpu.py tests/test_vae_gpu.py tests/test_ppo_gpu.py -q --timeout 900 -p no:cacheprovider "$@" > gpurun_out/t_$tag.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/t_$tag.log | head -30
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppo > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -c 300 gpurun_out/bench_$tag.err | grep -v amdgpu.ids
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
r = d["roofline"]
print("frames/s %.0f  ms/step %.3f  dominant %s %s frac %.4f" % (d["value"], d["ms_per_step"], r["kernel"], r["bound"], r["frac"]))
ops = d["per_op_ms"]
fam = {}
for k, v in ops.items():
    f = k.split(".")[-1] if "." in k else k
    fam[f] = fam.get(f, 0) + v
print({k: round(v, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})
print(list(ops.items())[:28])
PY
