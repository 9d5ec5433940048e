#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for dbg in 0 2 6 10 14 3; do echo "CFG=1 DBG=$dbg"; MI355_ARES_CFG=1 MI355_ARES_DBG=$dbg timeout 300 python tools/ares_bench.py 512 2>&1 | tail -2 | head -1 | cut -c1-120; done
