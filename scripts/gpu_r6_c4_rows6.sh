#!/bin/bash
# round 6: second pass of C4 -- chunks of consecutive workgroups per XCD (FENRIS_HIP_TWO_PASS_XCD_CHUNK = workgroups per chunk; a workgroup = 4 x npw nodes)
mkdir -p gpurun_out/r6_c4
V=""
for w in 4 16 64 256 1024 4096; do V="$V x$w:FENRIS_HIP_TWO_PASS_XCD_CHUNK=$w"; done
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 "prod:" $V "x64npw1:FENRIS_HIP_TWO_PASS_XCD_CHUNK=64,FENRIS_HIP_TWO_PASS_ROWS_GRID=1000000" "npw1:FENRIS_HIP_TWO_PASS_ROWS_GRID=1000000" 2>&1 | grep variant | tee gpurun_out/r6_c4/rows6_ab.txt
