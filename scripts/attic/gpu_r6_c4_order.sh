# round 6: C4 with K_e as upper node-block triangles -- the order in which the row gather walks the nodes
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c4
export TMPDIR=/tmp
timeout 600 python3 scripts/ab_in_context.py --config c4 --rounds 5 "tri_order0:FENRIS_HIP_TRI_ORDER=0" "tri_first_element:FENRIS_HIP_TRI_ORDER=1" "order0_grid16k:FENRIS_HIP_TWO_PASS_ROWS_GRID=16384" "first_grid16k:FENRIS_HIP_TRI_ORDER=1,FENRIS_HIP_TWO_PASS_ROWS_GRID=16384"  2>&1 | grep -v "amdgpu.ids" | tee $OUT/order2.txt
