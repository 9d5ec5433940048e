#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
MI355_ENC12_RING=1 MI355_ENC12_C2=1 timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "encoder_head_forward" > gpurun_out/call_s_tests.log 2>&1; echo "c2 op tests rc=$?"; tail -2 gpurun_out/call_s_tests.log
echo "--- op alone: old | ring | ring + c2"; python tools/enc12_ablate.py --only-product 2>&1 | grep -v amdgpu.ids | tail -1
MI355_ENC12_RING=1 python tools/enc12_ablate.py --only-product 2>&1 | grep -v amdgpu.ids | tail -1
MI355_ENC12_RING=1 MI355_ENC12_C2=1 python tools/enc12_ablate.py --only-product 2>&1 | grep -v amdgpu.ids | tail -1
STEPS=200 tools/ab_env.sh gpurun_out/r5_ab_c2.txt 4 "MI355_DEFAULTS=1" "MI355_ENC12_RING=1" "MI355_ENC12_RING=1 MI355_ENC12_C2=1"
