# round 6: C4 first pass -- wave-specialised form (hex27_roles.hpp: 4 matrix + 2 prologue wavefronts, double-buffered operands) against the
# alternating form (hex27_blocks.hpp), inside one context; the tests first
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c4
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_hex27_mfma.py tests/test_quadratic_elements.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
timeout 600 python3 scripts/ab_in_context.py --config c4 --rounds 5 "roles:" "alternating:FENRIS_HIP_EXP_HEX27_ROLES=0" "roles_1wg:FENRIS_HIP_HEX27_WGS_PER_CU=1" 2>&1 | grep -v "amdgpu.ids" | tee $OUT/roles.txt
bash scripts/gpu_kernel_split.sh c4 2>&1 | tee -a $OUT/roles.txt
