# usage: bash scripts/gpu_profile.sh [cells] [extra bench args]  -- rocprofv3 kernel stats + PMC passes of bench.py
CELLS=${1:-128}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="$GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cells $CELLS --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc1 -o run -- python3 $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/pmc2 -o run -- python3 $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o run -- python3 $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o run -- python3 $BENCH > $OUT/pmc4.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_ATOMIC -d $OUT/pmc5 -o run -- python3 $BENCH > $OUT/pmc5.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*.csv" | head -30
python3 scripts/summarize_prof.py gpurun_out/prof > gpurun_out/prof/summary.txt 2>&1
cat gpurun_out/prof/summary.txt
