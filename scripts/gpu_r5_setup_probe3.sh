mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_affine.py tests/test_hex8_rows.py tests/test_kernel_selection.py tests/test_hex27_mfma.py tests/test_gpu_parity.py tests/test_partition.py tests/test_distributed.py tests/test_full_size_slabs.py -x -q -m gpu 2>&1 | tail -5 > $OUT/tests3.txt
cat $OUT/tests3.txt
for cfg in ns ns-perturbed c5 c2; do
  FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/setup3_$cfg.txt
done
grep -h "context" $OUT/setup3_*.txt
bash scripts/gpu_r5_first_timeline.sh ns-perturbed > /dev/null 2>&1
