# rocprofv3 kernel stats of bench_more.py (Hex8 NeoHookean / StVK / per-point two-pass split)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_more
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/scripts/bench_more.py > $OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py gpurun_out/prof_more 2>&1 | head -24
