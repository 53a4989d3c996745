mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_affine.py tests/test_hex8_rows.py tests/test_kernel_selection.py tests/test_hex27_mfma.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > $OUT/tests.txt
for cfg in ns ns-perturbed c5 c4 c3; do
  FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/setup2_$cfg.txt
done
cat $OUT/tests.txt; grep -h "context\|cutting\|host copies" $OUT/setup2_*.txt
