#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "rwconv" 2>&1 | tail -3
python tools/op_bench.py deconv3.fwd conv2.dgrad
python tools/trace_rwconv.py deconv3.fwd | grep -v amdgpu
