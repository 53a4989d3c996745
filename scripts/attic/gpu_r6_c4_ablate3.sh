# round 6: stores of C4's first pass alone (FENRIS_HIP_ABLATE = 3) against the number of workgroups per CU (how much of K_e is in flight per L2)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c4
export TMPDIR=/tmp
V=""
for wg in 1 2 3 4 5; do for ab in 3 7; do V="$V wg${wg}_ab${ab}:FENRIS_HIP_HEX27_FORM=2,FENRIS_HIP_TRACE=1,FENRIS_HIP_ABLATE=$ab,FENRIS_HIP_HEX27_WGS_PER_CU=$wg"; done; done
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 $V 2>&1 | grep -v "amdgpu.ids\|trace\]" | tee $OUT/ablate3.txt
