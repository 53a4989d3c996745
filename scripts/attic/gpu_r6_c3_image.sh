# round 6: C3 -- the row image in LDS with a coalesced write-out (default) against stores straight from the lanes (FENRIS_HIP_TET4_DIRECT_STORES=1), one context
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c3; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c3
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_rows_kernel.py tests/test_rule_and_size_sweeps.py tests/test_high_valence.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tet or c3 or C3 or Tet or mask" 2>&1 | tail -3
V="image: direct:FENRIS_HIP_TET4_DIRECT_STORES=1 image_wgs3:FENRIS_HIP_PIPE_WGS_PER_CU=3 direct_wgs3:FENRIS_HIP_TET4_DIRECT_STORES=1,FENRIS_HIP_PIPE_WGS_PER_CU=3 image_nostores:FENRIS_HIP_ABLATE=4 image_noB:FENRIS_HIP_ABLATE=1"
timeout 900 python3 scripts/ab_in_context.py --config c3 --rounds 5 --reps 10 $V 2>&1 | grep -v "amdgpu.ids\|trace\]" | tee $OUT/image.txt
