#!/bin/bash
# round 6: the GPU suite with the two-pass launches forced onto small grids / other dealings (every two-pass test then runs many elements per workgroup,
# many nodes per wavefront, or one node per wavefront with XCD chunks of 32 nodes)
mkdir -p gpurun_out/r6_small
export TMPDIR=/tmp
(echo "== FENRIS_HIP_TWO_PASS_GRID=3 FENRIS_HIP_TWO_PASS_ROWS_GRID=2"
 FENRIS_HIP_TWO_PASS_GRID=3 FENRIS_HIP_TWO_PASS_ROWS_GRID=2 timeout 1500 python3 -m pytest tests -q -m gpu -k "not bench_launch and not full_size" 2>&1 | grep -v "HIP version\|ROCm version\|Hostname\|Librccl\|RCCL\|amdgpu.ids" | tail -3
 echo "== FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=1 FENRIS_HIP_TWO_PASS_XCD_CHUNK=8"
 FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=1 FENRIS_HIP_TWO_PASS_XCD_CHUNK=8 timeout 1500 python3 -m pytest tests -q -m gpu -k "not bench_launch and not full_size" 2>&1 | grep -v "HIP version\|ROCm version\|Hostname\|Librccl\|RCCL\|amdgpu.ids" | tail -3
 echo "== FENRIS_HIP_TWO_PASS_FULL=1 (full matrices between the passes of the generic form)"
 FENRIS_HIP_TWO_PASS_FULL=1 timeout 1500 python3 -m pytest tests -q -m gpu -k "not bench_launch and not full_size and not kernel_selection and not quadratic and not hex27_mfma and not tensor_operator" 2>&1 | grep -v "HIP version\|ROCm version\|Hostname\|Librccl\|RCCL\|amdgpu.ids" | tail -3) | grep -v "amdgpu.ids" | tee gpurun_out/r6_small/suite.txt
