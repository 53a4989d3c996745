#!/bin/bash
# round 6: the GPU suite with the affine / row-owner launches on one workgroup, on three, and on one workgroup per position (the ranges of the Laplace row
# loop then end behind every position: the loader's "behind the range" bit), and at loader depth 1
mkdir -p gpurun_out/r6_small
export TMPDIR=/tmp
F='HIP version\|ROCm version\|Hostname\|Librccl\|RCCL\|amdgpu.ids'
K="not bench_launch and not full_size"
(for g in 1 3 1000000; do
   echo "== FENRIS_HIP_AFFINE_GRID=$g FENRIS_HIP_PIPE_GRID=$g"
   FENRIS_HIP_AFFINE_GRID=$g FENRIS_HIP_PIPE_GRID=$g timeout 1500 python3 -m pytest tests -q -m gpu -k "$K" 2>&1 | grep -v "$F" | tail -2
 done
 echo "== FENRIS_HIP_AFFINE_DEPTH=1 FENRIS_HIP_AFFINE_GRID=5"
 FENRIS_HIP_AFFINE_DEPTH=1 FENRIS_HIP_AFFINE_GRID=5 timeout 1500 python3 -m pytest tests -q -m gpu -k "$K" 2>&1 | grep -v "$F" | tail -2) | tee gpurun_out/r6_small/suite_affine.txt
