#!/bin/bash
# round 6: second pass of C4 with the XCD chunks -- tests, what contiguous reads would still give (timing only), and where the reads are served from now
mkdir -p gpurun_out/r6_c4
timeout 900 python3 -m pytest tests/test_hex27_mfma.py tests/test_quadratic_elements.py tests/test_reproducible.py tests/test_gpu_parity.py tests/test_kernel_selection.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 "prod:" "contig:FENRIS_HIP_ABLATE=4096" "nostore:FENRIS_HIP_ABLATE=256" "noload:FENRIS_HIP_ABLATE=512" "nothing:FENRIS_HIP_ABLATE=3840" 2>&1 | grep variant | tee gpurun_out/r6_c4/rows8_ab.txt
bash scripts/gpu_pmc_mem.sh c4 xcd > /dev/null 2>&1
grep rows_from gpurun_out/pmcm_c4_xcd.txt | sed 's/void fenris_hip::k_rows_from_tri<unsigned char, false>//' | awk '{print $1, $3}'
