#!/bin/bash
# round 6: the second pass of C4 (k_rows_from_tri) taken apart -- FENRIS_HIP_ABLATE bits 0x100 no global stores, 0x200 no value loads,
# 0x400 no LDS adds, 0x800 no clearing (timing only; decimal in the environment; the first pass is the production one throughout)
mkdir -p gpurun_out/r6_c4
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 "prod:" "old:FENRIS_HIP_ROWS_TRI_OLD=1" "all:FENRIS_HIP_ABLATE=4096" "nostore:FENRIS_HIP_ABLATE=256" "noload:FENRIS_HIP_ABLATE=512" \
  "noadd:FENRIS_HIP_ABLATE=1024" "noclear:FENRIS_HIP_ABLATE=2048" "loadonly:FENRIS_HIP_ABLATE=3328" "storeonly:FENRIS_HIP_ABLATE=3584" "addonly:FENRIS_HIP_ABLATE=2816" "nothing:FENRIS_HIP_ABLATE=3840" \
  "noload_nostore:FENRIS_HIP_ABLATE=768" 2>&1 | grep variant | tee gpurun_out/r6_c4/rows_ablate.txt
