#!/bin/bash
# usage: tools/gpu_ab_lib.sh <libA.so> <libB.so> [reps]  -> interleaved bench runs of two builds of the library on the SAME box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --no-dp-form --steps 100"
for r in $(seq 1 ${3:-2}); do
for tag in A B; do
  if [ $tag = A ]; then L="$1"; else L="$2"; fi
  cp "$L" carla-ppo_amd/mi355/libmi355_carla.so
  timeout 300 python bench.py $X > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$tag.json").read().strip().splitlines()[-1])
    po=d["per_op_ms"]
    print("$tag [$L]", round(d["ms_per_step"],4), "ms |", " ".join("%s %.1f"%(k,po[k]*1e3) for k in ("conv2.dgrad","deconv3.fwd","deconv2.fwd","conv3.dgrad","deconv3.dgrad","conv2.fwd","conv3.fwd","deconv2.dgrad") if k in po))
except Exception as e:
    print("$tag ERR", e); print(open("gpurun_out/ab_$tag.err").read()[-500:])
PY
done; done
