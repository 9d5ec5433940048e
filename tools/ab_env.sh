#!/bin/bash
# usage (GPU box): tools/ab_env.sh <log> <rounds> "<VAR=val ...>" "<VAR=val ...>" ...   interleaved bench runs of environment-knob configurations on ONE box;
# prints every run and, per configuration, the minimum and the median ms/step (boxes and runs differ by 1-2 %: decisions are taken on the medians of >= 3 rounds)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
log=$1; rounds=$2; shift 2
# STEPS / WARMUP / EXTRA: bench.py arguments of every run (default 200 / 10; EXTRA e.g. "--precision bf16x3" or "--condition-ms 0")
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --no-dp-form --no-box --steps ${STEPS:-200} --warmup ${WARMUP:-10} ${EXTRA:-}"
rm -f "$log"
for r in $(seq 1 $rounds); do
  for cfg in "$@"; do
    ms=$(env $cfg timeout 300 python bench.py $X 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],4))" 2>/dev/null)
    echo "$cfg | $ms" >> "$log"
  done
done
python - "$log" <<'PY'
import sys, collections, statistics
d = collections.OrderedDict()
for ln in open(sys.argv[1]):
    k, v = ln.rsplit("|", 1)
    try: d.setdefault(k.strip(), []).append(float(v))
    except ValueError: d.setdefault(k.strip(), [])
for k, v in d.items():
    print("%-70s min %.4f  median %.4f  n=%d  %s" % (k, min(v) if v else float('nan'), statistics.median(v) if v else float('nan'), len(v), v))
PY
