#!/bin/bash
# round 6: C4 with the element's nodes in lexicographic order inside the two passes; wavefronts of a workgroup on interleaved / consecutive nodes
mkdir -p gpurun_out/r6_c4
timeout 1200 python3 -m pytest tests/test_hex27_mfma.py tests/test_quadratic_elements.py tests/test_reproducible.py tests/test_gpu_parity.py tests/test_kernel_selection.py tests/test_rule_and_size_sweeps.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 scripts/ab_in_context.py --config c4 --rounds 3 --reps 3 "lex:" "lex_inter:FENRIS_HIP_TWO_PASS_INTERLEAVE=1" "lex_inter2:FENRIS_HIP_TWO_PASS_INTERLEAVE=1,FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=2" "lex_npw2:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=2" "lex_npw1:FENRIS_HIP_TWO_PASS_NODES_PER_WAVE=1,FENRIS_HIP_TWO_PASS_ROWS_GRID=1000000" 2>&1 | grep variant | tee gpurun_out/r6_c4/lex_ab.txt
bash scripts/gpu_pmc_mem.sh c4 lex > /dev/null 2>&1
grep rows_from gpurun_out/pmcm_c4_lex.txt | sed 's/void fenris_hip::k_rows_from_tri<unsigned char, false>//' | awk '{print $1, $3}' | grep "RDREQ_sum\|READ_REQ_sum\|GUI\|HIT\|MISS"
