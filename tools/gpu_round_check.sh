#!/bin/bash
# usage (GPU box, one gpurun call): tools/gpu_round_check.sh <tag> [suite|nosuite] [profile|timeline|none]
#   the driver's GPU suite once (-x), the driver's exact bench command on the library at HEAD (-> gpurun_out/bench_<tag>_driver.json), the same command without
#   clock conditioning (two interleaved pairs, headline only: what the conditioning is worth on this box), then the round's profile passes or just the step timeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r05a}
if [ "${2:-suite}" = "suite" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/suite_$tag.log 2>&1
  echo "suite rc=$?"; tail -3 gpurun_out/suite_$tag.log
fi
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${tag}_driver.json 2> gpurun_out/bench_${tag}_driver.err
echo "bench rc=$?"
python - "$tag" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_%s_driver.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("driver cmd: %.1f frames/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))
print("box:", json.dumps(d.get("box")))
r = d["roofline"]; print("roofline:", r["kernel"][:40], r["avg_launch_ms"], r["frac"], r.get("frac_of_box"))
for k in ("bf16x3", "fp32", "mlp_vae", "ppo"):
    print(k, json.dumps({kk: vv for kk, vv in (d.get(k) or {}).items() if kk in ("frames_per_s", "ms_per_step", "ms_per_update", "error")}))
PY
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --no-dp-form --gpus 1 --steps 20 --warmup 5"
for r in 1 2; do
  for c in 0 300; do
    ms=$(timeout 300 python bench.py $X --condition-ms $c 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],4))" 2>/dev/null)
    echo "condition-ms $c | $ms" | tee -a gpurun_out/condition_ab_$tag.txt
  done
done
case "${3:-timeline}" in
  profile) tools/profile_round.sh $tag ;;
  timeline) tools/timeline.sh $tag; head -60 gpurun_out/timeline_$tag.md ;;
esac
