#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
X="--no-cpu-baseline --no-ppo --no-fp32 --no-replay"
timeout 300 python bench.py $X > $O/bench_eager.json 2> $O/bench_eager.err
timeout 300 python bench.py --graph 1 $X > $O/bench_graph.json 2> $O/bench_graph.err
MI355_BWD_STREAMS=0 timeout 300 python bench.py --graph 1 $X > $O/bench_1stream_graph.json 2> $O/bench_1stream_graph.err
for f in bench_eager bench_graph bench_1stream_graph; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), "frames/s", round(d["ms_per_step"],4), "ms", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"]*1e3,1), "us", d["roofline"]["bound"], round(d["roofline"]["frac"],3))
except Exception as e:
    print("$f", "ERR", e); print(open("$O/$f.err").read()[-600:])
PY
done
