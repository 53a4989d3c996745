mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_affine.py tests/test_hex8_rows.py tests/test_kernel_selection.py tests/test_hex27_mfma.py tests/test_gpu_parity.py tests/test_partition.py tests/test_distributed.py tests/test_full_size_slabs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
echo "--- default"; FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py ns-perturbed 2>&1 | grep -E "context|finished after|the rest"
echo "--- no lane tuning"; FENRIS_HIP_NO_LANE_TUNING=1 FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py ns-perturbed 2>&1 | grep -E "context|finished after|the rest"
echo "--- sleep 1 s between contexts"; FENRIS_SLEEP_BETWEEN=1 FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py ns-perturbed 2>&1 | grep -E "context|finished after|the rest"
echo "--- c5"; FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py c5 2>&1 | grep -E "context|finished after|set-up"
