#!/bin/bash
# round 6: second pass of C4 -- does its time follow the number of lines its value loads touch?  FENRIS_HIP_ABLATE 4096: every entry reads one contiguous
# run of 1 944 bytes (wrong values, timing only) instead of a run + up to 26 pieces of 72 bytes
mkdir -p gpurun_out/r6_c4
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 "prod:" "abl0:FENRIS_HIP_ABLATE=16384" "contig:FENRIS_HIP_ABLATE=4096" "loadonly:FENRIS_HIP_ABLATE=3328" "loadonly_contig:FENRIS_HIP_ABLATE=7424" \
   "nostore:FENRIS_HIP_ABLATE=256" "nostore_contig:FENRIS_HIP_ABLATE=4352" 2>&1 | grep variant | tee gpurun_out/r6_c4/rows4_ab.txt
