# round 6: what bounds C4's first pass -- phase ablation (instrumented instantiations: FENRIS_HIP_TRACE=1 + FENRIS_HIP_ABLATE bits 1 prologue, 2 matrix
# instructions, 4 stores) of the block form (FORM=2) and the tiles (FORM=0), inside one context; the second pass is in every figure
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c4
export TMPDIR=/tmp
V="prod_f2:FENRIS_HIP_HEX27_FORM=2 prod_f0:FENRIS_HIP_HEX27_FORM=0"
for f in 2 0; do for ab in 0 1 2 4 3 5 6 7; do V="$V f${f}_ab${ab}:FENRIS_HIP_HEX27_FORM=$f,FENRIS_HIP_TRACE=1,FENRIS_HIP_ABLATE=$ab"; done; done
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 $V 2>&1 | grep -v "amdgpu.ids\|trace\]" | tee $OUT/ablate.txt
