# round 6: C4 with K_e as upper node-block triangles -- the order in which the row gather walks the nodes (FENRIS_HIP_TRI_ORDER 0 index order,
# 1 + XCD-contiguous chunks, 2 Morton order + XCD chunks), against the tiles with full matrices
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c4
export TMPDIR=/tmp
timeout 600 python3 scripts/ab_in_context.py --config c4 --rounds 5 "tiles:FENRIS_HIP_HEX27_FORM=0" "tri_order0:FENRIS_HIP_TRI_ORDER=0" "tri_order1:FENRIS_HIP_TRI_ORDER=1" "tri_order2:FENRIS_HIP_TRI_ORDER=2" "tri_order3:FENRIS_HIP_TRI_ORDER=3" "tri_order3_grid4k:FENRIS_HIP_TRI_ORDER=3,FENRIS_HIP_TWO_PASS_ROWS_GRID=4096" "tri_order0_grid4k:FENRIS_HIP_TRI_ORDER=0,FENRIS_HIP_TWO_PASS_ROWS_GRID=4096" 2>&1 | grep -v "amdgpu.ids" | tee $OUT/order.txt
bash scripts/gpu_kernel_split.sh c4 2>&1 | tee -a $OUT/order.txt
