#!/bin/bash
# usage (GPU box): tools/gpu_call_b.sh <tag>   -- the new tests of the round first (fast feedback), the dense-layer tests, the MlpVAE step with the whole-grid XCD renumbering
# of its dense launches off / on (three interleaved pairs), then the whole suite with -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r05b}
timeout 900 python -m pytest tests/test_g_box_probe_gpu.py tests/test_h_script_traces_gpu.py "tests/test_d_c4_dp_gpu.py::test_dp_step_as_one_c_call_equals_the_host_loop" \
   "tests/test_ops_gpu.py::test_reparam_kl_fwd_bwd" -x -q -p no:cacheprovider > gpurun_out/new_tests_$tag.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/new_tests_$tag.log
for r in 1 2 3; do
  for v in 0 1; do
    echo "REMAP3=$v $(MI355_GEMM2_REMAP3=$v timeout 200 python tools/mlp_vae_bench.py --steps 100 --precision bf16 2>/dev/null | tail -1)" | tee -a gpurun_out/remap3_ab_$tag.txt
  done
done
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/suite_$tag.log 2>&1
echo "suite rc=$?"; tail -3 gpurun_out/suite_$tag.log
