# usage: bash scripts/gpu_pmc_mem.sh <config> <tag> [ENV=VALUE ...] -- where a configuration's reads are served from and what stalls them: L1 -> L2 requests,
# L2 -> fabric requests by size, the part that goes to DRAM, credit / tag stalls, address-translation and texture-addresser busy time
# (rocprofv3 --pmc passes, each its own run) into gpurun_out/pmcm_<config>_<tag>.txt
CFG=$1; TAG=$2; shift; shift
for kv in "$@"; do export "$kv"; done
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm_${CFG}_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="$GRAFT_REPO_ROOT/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-secondary --no-module-warmup --no-settle --placement-tries 0"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pmc$i -o run -- python3 $BENCH > $OUT/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py gpurun_out/pmcm_${CFG}_$TAG 2>&1 | grep -v "^== counters" | grep "k_rows_from\|k_hex27\|k_affine_rows<\|k_gather_rows\|k_hex8_rows\|k_spmv" | tee gpurun_out/pmcm_${CFG}_$TAG.txt
find gpurun_out/pmcm_${CFG}_$TAG -name "*.db" -delete
