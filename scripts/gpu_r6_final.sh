#!/bin/bash
# round 6, final state: the whole GPU suite, smoke(), the fuzzers, then the profile set
mkdir -p gpurun_out/r6f
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > gpurun_out/r6f/pytest_gpu.txt
tail -3 gpurun_out/r6f/pytest_gpu.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6f/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r6f/smoke.txt
timeout 1500 python3 scripts/fuzz_gather.py 1500 > gpurun_out/r6f/fuzz_gather.txt 2>&1; tail -1 gpurun_out/r6f/fuzz_gather.txt
timeout 900 python3 scripts/fuzz_vector.py 600 > gpurun_out/r6f/fuzz_vector.txt 2>&1; tail -1 gpurun_out/r6f/fuzz_vector.txt
timeout 600 python3 scripts/fuzz_pattern.py 300 > gpurun_out/r6f/fuzz_pattern.txt 2>&1; tail -1 gpurun_out/r6f/fuzz_pattern.txt
bash scripts/gpu_r6_profiles.sh
