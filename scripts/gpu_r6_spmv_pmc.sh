#!/bin/bash
# round 6: memory-path counters of the SpMV kernel (Hex8 elasticity 216^3, 19.7 GB of values)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_spmv
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pmc$i -o run -- python3 $GRAFT_REPO_ROOT/scripts/exp_spmv_forms.py > $OUT/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py gpurun_out/pmc_spmv 2>&1 | grep "k_spmv_blocked_half" | sed 's/void fenris_hip::k_spmv_blocked_half<3, 2>//' | awk '{print $1, $3, $4}' | tee gpurun_out/pmc_spmv.txt
find gpurun_out/pmc_spmv -name "*.db" -delete
