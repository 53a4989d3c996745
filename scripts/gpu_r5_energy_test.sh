export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_vector_tiles.py tests/test_vector_sweep.py tests/test_gpu_parity.py tests/test_compose.py tests/test_compact_table.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
python3 scripts/probe_energy_call.py 2>&1 | grep -v amdgpu
FENRIS_HIP_NO_MOMENT_RESIDUAL=1 python3 scripts/probe_energy_call.py 2>&1 | grep -v amdgpu
python3 scripts/fuzz_vector.py 300 2>&1 | tail -1
