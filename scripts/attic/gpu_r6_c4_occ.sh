# round 6: is the triangle row gather bound by occupancy?  LDS per workgroup scaled up (fewer workgroups per CU) and down (more: timing only, wrong sums)
export TMPDIR=/tmp
timeout 600 python3 scripts/ab_in_context.py --config c4 --rounds 3 "base:" "lds_x2:FENRIS_HIP_EXP_LDS_SCALE=2" "lds_div2:FENRIS_HIP_EXP_LDS_DIV=2" "lds_div4:FENRIS_HIP_EXP_LDS_DIV=4" 2>&1 | grep -v "amdgpu.ids"
