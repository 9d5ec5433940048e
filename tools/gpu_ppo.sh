#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_c_c3_ppo_gpu.py tests/test_e_c5_replay_gpu.py tests/test_ref_graph_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "^  File" | tail -25
for f in 0 1; do echo "MI355_PPO_FUSED=$f"; MI355_PPO_FUSED=$f python tools/ppo_probe.py; done
for f in 0 1; do echo "MI355_PPO_FUSED=$f M=2048"; MI355_PPO_FUSED=$f python tools/ppo_probe.py 2048; done
