#!/bin/bash
# rocprofv3 kernel stats of the C4 (Hex27 NeoHookean) configuration
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c4
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH_GATHER_ONLY=1 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py C4 > $OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py gpurun_out/prof_c4 2>&1 | head -12 | cut -c1-260
