# usage: bash scripts/gpu_pmc_quick.sh <config> <tag> [ENV=VALUE ...] -- FETCH_SIZE / WRITE_SIZE / L2 hit counters of the configuration's kernels (three rocprofv3
# --pmc passes, each its own run) into gpurun_out/pmcq_<config>_<tag>.txt
CFG=$1; TAG=$2; shift; shift
for kv in "$@"; do export "$kv"; done
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_${CFG}_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="$GRAFT_REPO_ROOT/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-secondary --no-module-warmup --no-settle --placement-tries 0"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc1 -o run -- python3 $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc2 -o run -- python3 $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $OUT/pmc3 -o run -- python3 $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/pmc4 -o run -- python3 $BENCH > $OUT/pmc4.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py gpurun_out/pmcq_${CFG}_$TAG 2>&1 | grep -v "^== counters" | tee gpurun_out/pmcq_${CFG}_$TAG.txt
find gpurun_out/pmcq_${CFG}_$TAG -name "*.db" -delete
