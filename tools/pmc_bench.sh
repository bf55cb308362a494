#!/bin/bash
# usage: tools/pmc_bench.sh <kernel-name-regex> ; PMC counters (own passes) of matching kernels over 2 bench steps
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pmc
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc -o p -- python /root/repo/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-kernel-timing > /tmp/pmc_run.log 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmc 40 all 2>&1 | grep -A1 -E "$1"
done
