#!/bin/bash
# usage: tools/pmc_layers.sh <layers> ; collects PMC counters for the selected layers' launches (own passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
export ONLY=$1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  rm -rf /tmp/pmc
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc -o p -- python /root/repo/tools/layer_times.py > /tmp/pmc_run.log 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmc 6 2>&1 | grep -A1 -E "conv_halo|conv_mfma_kernel|wgrad_mfma|wgrad_halo" 
done
