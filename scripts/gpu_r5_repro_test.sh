export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_reproducible.py tests/test_kernel_selection.py tests/test_bindings.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30
