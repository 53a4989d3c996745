# round 6: C3 (Tet4 row-owner kernel) -- phase ablation inside one context (FENRIS_HIP_ABLATE: 1 no phase B, 2 no products, 4 no global stores,
# 16 every lane's block as one contiguous 72-byte run next to its neighbour's: what ideal stores would cost -- timing only), workgroups per CU
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c3; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c3
export TMPDIR=/tmp
V="base:"
for ab in 1 2 4 16 3 5 6 7 19; do V="$V ab${ab}:FENRIS_HIP_ABLATE=$ab"; done
for wg in 1 2 3 4; do V="$V wgs${wg}:FENRIS_HIP_PIPE_WGS_PER_CU=$wg"; done
timeout 900 python3 scripts/ab_in_context.py --config c3 --rounds 5 --reps 10 $V 2>&1 | grep -v "amdgpu.ids\|trace\]" | tee $OUT/ablate.txt
