#!/bin/bash
# usage: tools/gpu_tests.sh [pytest args]   -> gpurun_out/pytest_last.log (whole -m gpu suite by default)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/pytest_last.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_last.log
grep -v "^  File\|Extension modules" gpurun_out/pytest_last.log | tail -40
