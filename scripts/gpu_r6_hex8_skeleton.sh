# round 6: k_hex8_rows -- what the loader / store skeleton is bound by: FENRIS_HIP_ABLATE bit 1 no global stores, 2 no phase C, 4 no phase B, 16 no global loads in the sweep
export TMPDIR=/tmp
V="dbg:FENRIS_HIP_ABLATE=64 nostores:FENRIS_HIP_ABLATE=65 noB_noC:FENRIS_HIP_ABLATE=70 noB_noC_nostores:FENRIS_HIP_ABLATE=71 noB_noC_noloads:FENRIS_HIP_ABLATE=86 noB_noC_neither:FENRIS_HIP_ABLATE=87 wgs1:FENRIS_HIP_PIPE_WGS_PER_CU=1"
timeout 900 python3 scripts/ab_in_context.py --config ns-perturbed --rounds 3 --reps 3 $V 2>&1 | grep -v "amdgpu.ids\|trace\]"
