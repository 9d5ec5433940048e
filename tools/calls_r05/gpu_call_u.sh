#!/bin/bash
# one gpurun call: the fallback forms still pass -- the fused encoder head off, its compiler-scheduled forms, the round's sequencing knobs off
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
T="tests/test_vae_gpu.py tests/test_a_c2_b512_gpu.py tests/test_b_c1_epoch_gpu.py"
for cfg in "MI355_ENC12=0" "MI355_ENC12_RING=0 MI355_ENC12_C2=0" "MI355_ENC12_C2=0" "MI355_MID_FLUSH=0 MI355_FIN_SIDE=0 MI355_ADAM_FRAG=0 MI355_LATENT_SPLIT=32"; do
  env $cfg timeout 900 python -m pytest $T tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "not mlp" > gpurun_out/call_u.log 2>&1; echo "$cfg rc=$? $(grep -E 'passed|failed' gpurun_out/call_u.log | tail -1)"
done
