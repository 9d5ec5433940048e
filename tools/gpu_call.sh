#!/bin/bash
# ONE parametrised `gpurun` call (replaces the sixteen one-off tools/calls_r05/gpu_call_*.sh of round 5; what each of those ran is in git history and in
# profiles/r05_*): a tag and a sequence of actions separated by `--`, executed in order on one box against one library; everything lands in gpurun_out/*_<tag>.*
#
#   gpurun --timeout 1800 -- 'tools/gpu_call.sh r06x tests "tests/test_ops_gpu.py -k filter_gradient" -- ab 3 "MI355_TW_LDEC=0" "MI355_TW_LDEC=1" -- suite 2 -- driver -- profile'
#
# actions
#   tests "<pytest arguments>"          targeted tests (-x -q), tail of the log; quote -k expressions with escaped or single quotes inside: tests "tests/test_ops_gpu.py -k 'a or b'" 
#   suite [n]                           the driver's GPU suite (pytest -m gpu) n times in a row (default 1)
#   smoke                               __graft_entry__.smoke()
#   ab <rounds> "<VAR=val ...>" ...     interleaved step-time A/B of environment-knob configurations (tools/ab_env.sh; STEPS / WARMUP / EXTRA pass through)
#   perop <regex> "<VAR=val ...>" ...   isolated per-op times of the matching ops under each configuration (tools/ab_perop.sh)
#   driver                              the driver's exact bench command (--gpus 1 --steps 20 --warmup 5) -> bench_<tag>_driver.json + a one-screen digest
#   profile                             the round's bench + rocprofv3 kernel trace + three PMC passes + timeline (tools/profile_round.sh <tag>)
#   run "<shell command>"               anything else (ablation tools, probes), output to run_<tag>_<k>.log and the screen
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1; shift
k=0
while [ $# -gt 0 ]; do
  act=$1; shift
  args=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
  [ "$1" = "--" ] && shift
  k=$((k + 1))
  echo "=== [$tag] $act ${args[*]}"
  case $act in
    tests) eval "timeout 1500 python -m pytest ${args[0]} -x -q -p no:cacheprovider" > gpurun_out/tests_${tag}_$k.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/tests_${tag}_$k.log ;;      # (eval: quoted -k expressions inside the argument string survive)
    suite) for i in $(seq 1 ${args[0]:-1}); do timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/suite_${tag}_$i.log 2>&1; echo "run $i rc=$?"; grep -E "passed|failed|error" gpurun_out/suite_${tag}_$i.log | tail -2; done ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_$tag.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/smoke_$tag.log ;;
    ab) r=${args[0]}; tools/ab_env.sh gpurun_out/ab_${tag}_$k.txt $r "${args[@]:1}" | tail -$(( ${#args[@]} - 1 )) ;;
    perop) pat=${args[0]}; tools/ab_perop.sh gpurun_out/perop_${tag}_$k.txt "$pat" "${args[@]:1}" ;;
    driver) timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${tag}_driver.json 2> gpurun_out/bench_${tag}_driver.err; echo "rc=$?"
            python - "$tag" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_%s_driver.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("driver cmd: %.1f frames/s  %.4f ms/step; value_200 %s; library %s" % (d["value"], d["ms_per_step"], (d.get("value_200") or {}).get("ms_per_step"), d["config"]["library"]))
print("box:", {k: v for k, v in (d.get("box") or {}).items() if k != "how"})
r = d["roofline"]; print("roofline:", r["kernel"][:50], "in-step ms", round(r["avg_launch_ms"], 4), "frac", round(r["frac"], 3), "of box", round(r.get("frac_of_box", 0), 3), "step_traffic", r.get("step_traffic"))
print("in-step candidates:", d.get("per_op_in_step_ms"))
for k in ("bf16x3", "fp32", "mlp_vae", "ppo", "replay"):
    print(k, {kk: vv for kk, vv in (d.get(k) or {}).items() if kk in ("frames_per_s", "ms_per_step", "ms_per_update", "error", "roofline", "encode_roofline", "seconds")})
PY
            ;;
    profile) tools/profile_round.sh $tag ;;
    run) bash -c "${args[0]}" 2>&1 | tee gpurun_out/run_${tag}_$k.log | tail -60 ;;
    *) echo "unknown action $act" ;;
  esac
done
