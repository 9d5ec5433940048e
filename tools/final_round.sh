#!/bin/bash
# one gpurun call = one box, one library: the driver's GPU suite RUNS times in a row (default 5), smoke(), the driver's exact bench command on the library at HEAD
# (-> gpurun_out/bench_<tag>_driver.json), then the round's bench + profile passes (tools/profile_round.sh) unless the second argument is noprofile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r05a}
: > gpurun_out/suite_loop_$tag.log
for i in $(seq 1 ${RUNS:-5}); do
  echo "=== run $i ===" >> gpurun_out/suite_loop_$tag.log
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/suite_run_${tag}_$i.log 2>&1
  echo "rc=$?" >> gpurun_out/suite_loop_$tag.log
  grep -E "passed|failed|error" gpurun_out/suite_run_${tag}_$i.log | tail -3 >> gpurun_out/suite_loop_$tag.log
done
cat gpurun_out/suite_loop_$tag.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_$tag.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_$tag.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${tag}_driver.json 2> gpurun_out/bench_${tag}_driver.err; echo "driver bench rc=$?"
python - "$tag" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_%s_driver.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("driver cmd: %.1f frames/s  %.4f ms/step; library %s" % (d["value"], d["ms_per_step"], d["config"]["library"]))
print("box:", {k: v for k, v in (d.get("box") or {}).items() if k != "how"})
r = d["roofline"]; print("roofline:", r["kernel"][:50], "in-step ms", round(r["avg_launch_ms"], 4), "frac", round(r["frac"], 3), "of box", round(r.get("frac_of_box", 0), 3))
print("in-step candidates:", d.get("per_op_in_step_ms"))
for k in ("bf16x3", "fp32", "mlp_vae", "ppo"):
    print(k, {kk: vv for kk, vv in (d.get(k) or {}).items() if kk in ("frames_per_s", "ms_per_step", "ms_per_update", "error")})
PY
[ "$2" = "noprofile" ] || tools/profile_round.sh $tag
