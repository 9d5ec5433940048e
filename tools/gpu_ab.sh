#!/bin/bash
# usage: tools/gpu_ab.sh "ENV_A" "ENV_B" [reps]   -> interleaved bench runs of two environments on the SAME box (boxes differ by up to 20 %)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --steps 100"
for r in $(seq 1 ${3:-2}); do
for tag in A B; do
  if [ $tag = A ]; then E="$1"; else E="$2"; fi
  env $E timeout 300 python bench.py $X > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$tag.json").read().strip().splitlines()[-1])
    po=d["per_op_ms"]
    print("$tag [$E]", round(d["ms_per_step"],4), "ms |", " ".join("%s %.1f"%(k,po[k]*1e3) for k in ("conv1.fwd","conv2.dgrad","deconv3.fwd","deconv4.dgrad","deconv3.wgrad","conv2.fwd","deconv3.dgrad","deconv4.fwd","conv1.wgrad") if k in po))
except Exception as e:
    print("$tag ERR", e); print(open("gpurun_out/ab_$tag.err").read()[-500:])
PY
done; done
