mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5res; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5res
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_vector_tiles.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30 > $OUT/tests_res2.txt
tail -15 $OUT/tests_res2.txt
bash scripts/gpu_r5_residual_prof.sh > /dev/null 2>&1
head -3 $OUT/summary.txt | cut -c1-220; grep "INSTS_VALU\|ACTIVE_INST_VALU\|GRBM" $OUT/summary.txt | head -6
