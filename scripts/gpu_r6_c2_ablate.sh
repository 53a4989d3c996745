# round 6: C2 (Hex8 Poisson 128^3: k_affine_records + k_affine_rows<Laplace>) -- phase ablation inside one context.  FENRIS_HIP_ABLATE bits of the
# instrumented instantiation: 1 no global stores, 2 no products, 4 no record fetches, 16 nothing off
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c2; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c2
export TMPDIR=/tmp
V="prod:"
for ab in 16 1 2 4 3 5 6 7; do V="$V ab${ab}:FENRIS_HIP_ABLATE=$ab"; done
for wg in 2 3 4 5 6; do V="$V wgs${wg}:FENRIS_HIP_AFFINE_WGS_PER_CU=$wg"; done
timeout 900 python3 scripts/ab_in_context.py --config c2 --rounds 5 --reps 20 $V 2>&1 | grep -v "amdgpu.ids\|trace\]" | tee $OUT/ablate.txt
