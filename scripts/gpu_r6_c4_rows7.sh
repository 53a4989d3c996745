#!/bin/bash
# round 6: second pass of C4 -- nodes per wavefront x XCD chunk (FENRIS_HIP_TWO_PASS_XCD_CHUNK = nodes per wavefront-slot of a chunk: chunk = 4 x that many nodes)
mkdir -p gpurun_out/r6_c4
V=""
for n in 2 4 8; do for w in 256 1024 4096; do V="$V n${n}x$w:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=$n,FENRIS_HIP_TWO_PASS_XCD_CHUNK=$w"; done; done
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 "prod:" "old:FENRIS_HIP_ROWS_TRI_OLD=1" "tstore:FENRIS_HIP_ABLATE=16384" $V 2>&1 | grep variant | tee gpurun_out/r6_c4/rows7_ab.txt
