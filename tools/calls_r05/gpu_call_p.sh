#!/bin/bash
# one gpurun call: the ring form of the fused encoder head's frame loads -- parity (op tests), the op alone, the step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
MI355_ENC12_RING=1 timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "encoder_head_forward" > gpurun_out/call_p_tests.log 2>&1; echo "ring op tests rc=$?"; tail -2 gpurun_out/call_p_tests.log
MI355_ENC12_RING=1 timeout 600 python -m pytest tests/test_vae_gpu.py tests/test_a_c2_vae_gpu.py -q -x -p no:cacheprovider > gpurun_out/call_p_tests2.log 2>&1; echo "ring engine tests rc=$?"; tail -2 gpurun_out/call_p_tests2.log
echo "--- op alone, ring off"; python tools/enc12_ablate.py --only-product 2>&1 | grep -v amdgpu.ids
echo "--- op alone, ring on"; MI355_ENC12_RING=1 python tools/enc12_ablate.py --only-product 2>&1 | grep -v amdgpu.ids
STEPS=200 tools/ab_env.sh gpurun_out/r5_ab_ring.txt 4 "MI355_DEFAULTS=1" "MI355_ENC12_RING=1"
