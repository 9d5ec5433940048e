#!/bin/bash
# usage (GPU box): tools/ab_perop.sh <log> <pattern> "<VAR=val ...>" "<VAR=val ...>" ...   ISOLATED per-op times (bench.py per_op_ms: every op bracketed by HIP events, one at a
# time) of the ops whose name matches <pattern> under several environment-knob configurations, next to ms/step -- which kernel a step-level A/B moved
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
log=$1; pat=$2; shift 2
X="--no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --no-dp-form --no-box --steps ${STEPS:-100} --warmup ${WARMUP:-10}"
rm -f "$log"
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py $X 2>/dev/null | python -c "
import sys, json, re
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ops = {k: v for k, v in d['per_op_ms'].items() if re.search(r'$pat', k)}
print('$cfg | ms/step %.4f | ' % d['ms_per_step'] + '  '.join('%s %.1f' % (k, 1e3 * v) for k, v in sorted(ops.items())) + ' | sum %.1f us' % (1e3 * sum(ops.values())))
" | tee -a "$log"
done
