#!/bin/bash
# round 6: the generic two-pass form with lower node-block triangles between the passes (3 x 3 blocks on 3D elements) -- tests, then Hex8 / Tet4 timings
# against the full column-major matrices (FENRIS_HIP_TWO_PASS_FULL=1) on the same box
mkdir -p gpurun_out/r6_tri
#timeout 1500 python3 -m pytest tests/test_hex27_mfma.py tests/test_quadratic_elements.py tests/test_reproducible.py tests/test_gpu_parity.py tests/test_kernel_selection.py tests/test_rule_and_size_sweeps.py tests/test_tensor_operator.py tests/test_high_valence.py tests/test_patch.py -x -q -m gpu 2>&1 | tail -4
(echo "triangles"; timeout 600 python3 scripts/bench_hex8_nh.py 128 2>&1 | grep operator | head -2
 echo "FENRIS_HIP_TWO_PASS_FULL=1"; FENRIS_HIP_TWO_PASS_FULL=1 timeout 600 python3 scripts/bench_hex8_nh.py 128 2>&1 | grep operator | head -2
 echo "triangles"; timeout 600 python3 scripts/bench_hex8_nh.py 216 2>&1 | grep operator | head -2
 echo "triangles, Tet4 NeoHookean"; timeout 600 python3 scripts/bench_tet_nh.py 2>&1 | tail -3
 echo "full, Tet4 NeoHookean"; FENRIS_HIP_TWO_PASS_FULL=1 timeout 600 python3 scripts/bench_tet_nh.py 2>&1 | tail -3) | tee gpurun_out/r6_tri/hex8_nh.txt
