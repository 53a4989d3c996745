# round 6: the stores of C4's first pass alone (prologue and matrix instructions off: FENRIS_HIP_ABLATE = 3), all / direct only (+ 8) / mirrored only (+ 16) / none (+ 24)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c4
export TMPDIR=/tmp
V=""
for ab in 3 11 19 27 0 8 16 24; do V="$V f2_ab${ab}:FENRIS_HIP_HEX27_FORM=2,FENRIS_HIP_TRACE=1,FENRIS_HIP_ABLATE=$ab"; done
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 $V 2>&1 | grep -v "amdgpu.ids\|trace\]" | tee $OUT/ablate2.txt
