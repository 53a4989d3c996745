#!/bin/bash
# round 6: second pass of C4, persistent wavefronts over interleaved nodes (FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=0) at several grids
mkdir -p gpurun_out/r6_c4
V=""
for g in 1024 2048 4096 8192 16384 32768 65536; do V="$V g$g:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=0,FENRIS_HIP_TWO_PASS_ROWS_GRID=$g"; done
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 "npw1:" "old:FENRIS_HIP_ROWS_TRI_OLD=1" $V "g4096nothing:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=0,FENRIS_HIP_TWO_PASS_ROWS_GRID=4096,FENRIS_HIP_ABLATE=3840" 2>&1 | grep variant | tee gpurun_out/r6_c4/rows3_ab.txt
