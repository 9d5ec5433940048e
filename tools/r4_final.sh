#!/bin/bash
# one gpurun call: the driver's GPU suite five times in a row, then the round's bench + profile passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r04a}
: > gpurun_out/suite_loop_$tag.log
for i in $(seq 1 ${RUNS:-5}); do
  echo "=== run $i ===" >> gpurun_out/suite_loop_$tag.log
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/suite_run_${tag}_$i.log 2>&1
  echo "rc=$?" >> gpurun_out/suite_loop_$tag.log
  grep -E "passed|failed|error" gpurun_out/suite_run_${tag}_$i.log | tail -3 >> gpurun_out/suite_loop_$tag.log
done
cat gpurun_out/suite_loop_$tag.log
[ "$2" = "noprofile" ] || tools/profile_round.sh $tag
