#!/bin/bash
# round 6: second pass of C4 -- non-temporal value loads (FENRIS_HIP_ABLATE 8192) / row stores (16384)
mkdir -p gpurun_out/r6_c4
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 "prod:" "abl0:FENRIS_HIP_ABLATE=32768" "ntload:FENRIS_HIP_ABLATE=8192" "ntstore:FENRIS_HIP_ABLATE=16384" "ntboth:FENRIS_HIP_ABLATE=24576" 2>&1 | grep variant | tee gpurun_out/r6_c4/rows5_ab.txt
