mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5res; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5res
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_vector_tiles.py tests/test_vector_sweep.py tests/test_source.py tests/test_gpu_parity.py tests/test_affine.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30 > $OUT/tests_res.txt
tail -5 $OUT/tests_res.txt
python3 scripts/time_residual.py 2>&1 | grep residual
FENRIS_HIP_NO_MOMENT_RESIDUAL=1 python3 scripts/time_residual.py 2>&1 | grep residual
python3 scripts/time_residual.py 2>&1 | grep residual
python3 scripts/fuzz_vector.py 300 2>&1 | tail -3
