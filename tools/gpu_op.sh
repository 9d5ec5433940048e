#!/bin/bash
# usage: tools/gpu_op.sh "<op_bench args>" [pmc]   -> op timings (+ one PMC pass of the same run when the 2nd arg is "pmc")
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/op_bench.py $1 2>&1 | tee gpurun_out/op_last.log
if [ "$2" = "pmc" ]; then
  cd /tmp
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/pmc_op -o p -- python $GRAFT_REPO_ROOT/tools/op_bench.py $1 --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc_op.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $GRAFT_REPO_ROOT/gpurun_out/pmc_op2 -o p -- python $GRAFT_REPO_ROOT/tools/op_bench.py $1 --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc_op2.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/rocpd_pmc_summary.py $(find gpurun_out/pmc_op -name "*.db" | head -1) 2>&1 | grep -v "at::\|rocprim" | tee gpurun_out/pmc_op.md
  python tools/rocpd_pmc_summary.py $(find gpurun_out/pmc_op2 -name "*.db" | head -1) 2>&1 | grep -v "at::\|rocprim" | tee gpurun_out/pmc_op2.md
  rm -rf gpurun_out/pmc_op gpurun_out/pmc_op2
fi
