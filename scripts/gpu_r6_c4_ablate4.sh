# round 6: C4 first pass (triangle form), what ideal stores would buy: FENRIS_HIP_ABLATE bit 8 = the same bytes as whole 1 KB runs per store instruction (wrong
# places: timing only), with everything else on (8), stores only (11)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c4
export TMPDIR=/tmp
V="prod:"
for ab in 0 8 3 11 7 4; do V="$V ab${ab}:FENRIS_HIP_TRACE=1,FENRIS_HIP_ABLATE=$ab"; done
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 $V 2>&1 | grep -v "amdgpu.ids\|trace\]" | tee $OUT/ablate4.txt
