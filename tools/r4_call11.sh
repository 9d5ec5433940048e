#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for cfg in 1 0; do for dbg in 0 1 2 3; do echo "CFG=$cfg DBG=$dbg"; MI355_ARES_CFG=$cfg MI355_ARES_DBG=$dbg timeout 300 python tools/ares_bench.py 512 2>&1 | tail -2 | cut -c1-200; done; done
