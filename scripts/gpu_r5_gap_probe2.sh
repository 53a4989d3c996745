mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_affine.py tests/test_hex8_rows.py tests/test_kernel_selection.py tests/test_hex27_mfma.py tests/test_gpu_parity.py tests/test_partition.py tests/test_distributed.py tests/test_full_size_slabs.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -60 > $OUT/tests4.txt
echo "--- default"; FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py ns-perturbed 2>&1 | grep -E "context|finished after|the rest"
echo "--- malloc never returns memory"; MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=1000000000000 FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py ns-perturbed 2>&1 | grep -E "context|finished after|the rest"
