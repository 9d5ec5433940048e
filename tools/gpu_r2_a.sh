#!/bin/bash
# round 2, GPU call A: whole -m gpu suite, bench (eager / graph / fp32 frame table A-B), kernel-trace stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x --deselect tests/test_vae_gpu.py::test_data_parallel_two_ranks_on_the_gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_eager.json 2> $O/bench_eager.err; echo "bench rc=$?"
timeout 300 python bench.py --graph 1 --no-cpu-baseline --no-ppo --no-fp32 --no-replay > $O/bench_graph.json 2> $O/bench_graph.err; echo "graph rc=$?"
timeout 300 python bench.py --frames f32 --no-cpu-baseline --no-ppo --no-fp32 --no-replay > $O/bench_f32frames.json 2> $O/bench_f32frames.err; echo "f32frames rc=$?"
MI355_BWD_STREAMS=0 timeout 300 python bench.py --no-cpu-baseline --no-ppo --no-fp32 --no-replay > $O/bench_1stream.json 2> $O/bench_1stream.err; echo "1stream rc=$?"
MI355_BWD_STREAMS=0 timeout 300 python bench.py --graph 1 --no-cpu-baseline --no-ppo --no-fp32 --no-replay > $O/bench_1stream_graph.json 2> $O/bench_1stream_graph.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-fp32 --no-replay > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/prof_summary.md 2>&1 || true
rm -rf $O/prof
tools/timeline.sh r2a; mv gpurun_out/timeline_r2a.md $O/ 2>/dev/null
for f in bench_eager bench_graph bench_f32frames bench_1stream bench_1stream_graph; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), "frames/s", round(d["ms_per_step"],4), "ms", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"]*1e3,1), "us", d["roofline"]["bound"], round(d["roofline"]["frac"],3))
except Exception as e:
    print("$f", "ERR", e)
PY
done
