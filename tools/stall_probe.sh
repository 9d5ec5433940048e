#!/bin/bash
# usage (GPU box): tools/stall_probe.sh <tag>  -- does THIS box stall the latent chain of the backward pass behind the filter-gradient queue (dense1.dgrad = tallk_kernel<2>: 13-14 us
# in the step on most boxes, 57-65 us on some: DESIGN_HISTORY 3.16e)?  If it does, A/B the knobs aimed at it on this very box.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
tag=$1
tools/timeline.sh ${tag}_0
d=$(grep -E 'tallk_kernel<2>' gpurun_out/timeline_${tag}_0.md | head -1 | cut -d'|' -f5 | tr -d ' ')
echo "box: $(grep -m1 'step wall' gpurun_out/timeline_${tag}_0.md); dense1.dgrad in the step: $d us"
if python -c "import sys; sys.exit(0 if float('$d') > 30 else 1)"; then
  echo "STALL BOX"
  MI355_MID_FLUSH_LATE=1 tools/timeline.sh ${tag}_1
  echo "late submit: $(grep -m1 'step wall' gpurun_out/timeline_${tag}_1.md); dense1.dgrad $(grep -E 'tallk_kernel<2>' gpurun_out/timeline_${tag}_1.md | head -1 | cut -d'|' -f5) us"
  STEPS=200 tools/ab_env.sh gpurun_out/ab_${tag}.txt 4 "MI355_MID_FLUSH_LATE=0" "MI355_MID_FLUSH_LATE=1" "MI355_MID_FLUSH=0" | tail -3
fi
