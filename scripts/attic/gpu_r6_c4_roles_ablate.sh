# round 6: the wave-specialised first pass of C4 (hex27_roles.hpp) taken apart: FENRIS_HIP_ABLATE 1 no chain / P2, 2 no matrix instructions, 4 no stores
export TMPDIR=/tmp
V="roles:"
for ab in 1 2 4 3 5 6 7; do V="$V ab$ab:FENRIS_HIP_ABLATE=$ab"; done
timeout 600 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 $V 2>&1 | grep -v "amdgpu.ids"
