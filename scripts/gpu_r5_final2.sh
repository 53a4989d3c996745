#!/bin/bash
mkdir -p gpurun_out/r5f
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > gpurun_out/r5f/pytest_gpu2.txt
tail -2 gpurun_out/r5f/pytest_gpu2.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5f/smoke2.txt 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5f/smoke2.txt
python3 bench.py > gpurun_out/r5f/bench_default2.json 2> gpurun_out/r5f/bench_default2.err; echo "bench rc=$?"
python3 scripts/bench_other_kernels.py > gpurun_out/r5f/other_kernels2.jsonl 2> gpurun_out/r5f/other2.err; echo "other rc=$?"
