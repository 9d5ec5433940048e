#!/bin/bash
# PMC passes on the ares microbench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
pass() {
  tag=$1; shift
  timeout 200 rocprofv3 --pmc $@ -d $R/gpurun_out/pmc_$tag -o p -- python $R/tools/ares_bench.py 512 > $R/gpurun_out/pmc_$tag.log 2>&1
  python $R/tools/rocpd_pmc_summary.py $(find $R/gpurun_out/pmc_$tag -name "*.db" | head -1) ares > $R/gpurun_out/pmc_$tag.md 2>> $R/gpurun_out/pmc_$tag.log
  rm -rf $R/gpurun_out/pmc_$tag
  cat $R/gpurun_out/pmc_$tag.md
}
pass ares_sq GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS
pass ares_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass ares_fetch FETCH_SIZE
pass ares_l2 TCC_HIT_sum TCC_MISS_sum
