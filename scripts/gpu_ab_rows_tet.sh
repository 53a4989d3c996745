#!/bin/bash
# C3 (Tet4 elasticity, BCC res 75, permuted): pipelined kernel against the row-owner kernel (FENRIS_HIP_ROWS=1)
run() { BENCH_GATHER_ONLY=1 python scripts/bench_configs.py C3 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); m=d['modes']['gather']; print('$1', m['kernel'], round(m['kernel_ms'],3), 'finite', m['finite'])"; }
run pipelined
FENRIS_HIP_ROWS=1 run rows_2wg
FENRIS_HIP_ROWS=1 FENRIS_HIP_PIPE_WGS_PER_CU=3 run rows_3wg
FENRIS_HIP_ROWS=1 FENRIS_HIP_PIPE_WGS_PER_CU=4 run rows_4wg
run pipelined
