#!/bin/bash
# usage: tools/gpu_narrow.sh "<narrow_bench args>" [pmc]  -> isolated timings of the narrow-layer ops (+ two PMC passes of the same ops when the 2nd arg is "pmc")
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/narrow_bench.py $1 2>&1 | tee gpurun_out/narrow_last.log
if [ "$2" = "pmc" ]; then
  cd /tmp
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR -d $R/gpurun_out/pmc_n1 -o p -- python $R/tools/narrow_bench.py $1 --iters 3 > $R/gpurun_out/pmc_n1.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmc_n2 -o p -- python $R/tools/narrow_bench.py $1 --iters 3 > $R/gpurun_out/pmc_n2.log 2>&1
  timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d $R/gpurun_out/pmc_n3 -o p -- python $R/tools/narrow_bench.py $1 --iters 3 > $R/gpurun_out/pmc_n3.log 2>&1
  cd $R
  for n in 1 2 3; do python tools/rocpd_pmc_summary.py $(find gpurun_out/pmc_n$n -name "*.db" | head -1) 2>&1 | grep -v "at::\|rocprim\|elementwise\|Philox\|distribution" | tee gpurun_out/pmc_n$n.md; tail -3 gpurun_out/pmc_n$n.log | grep -i "error\|invalid" ; done
  rm -rf gpurun_out/pmc_n1 gpurun_out/pmc_n2 gpurun_out/pmc_n3
fi
