#!/bin/bash
# round 6: second pass of C4 -- node metadata requested two nodes ahead, one position load per entry, no padded entries -- against the previous form
# (in one context), then taken apart (FENRIS_HIP_ABLATE, decimal: 256 no global stores, 512 no value loads, 1024 no LDS adds, 2048 no clearing)
mkdir -p gpurun_out/r6_c4
timeout 900 python3 -m pytest tests/test_hex27_mfma.py tests/test_quadratic_elements.py tests/test_reproducible.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 "new:" "old:FENRIS_HIP_ROWS_TRI_OLD=1" "npw1:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=1" "npw4:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=4" "npw8:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=8" "npw32:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=32" "npw64:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=64" "npw400:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=400" "nostore:FENRIS_HIP_ABLATE=256" "noload:FENRIS_HIP_ABLATE=512" \
  "noadd:FENRIS_HIP_ABLATE=1024" "loadonly:FENRIS_HIP_ABLATE=3328" "storeonly:FENRIS_HIP_ABLATE=3584" "nothing:FENRIS_HIP_ABLATE=3840" 2>&1 | grep variant | tee gpurun_out/r6_c4/rows2_ab.txt
