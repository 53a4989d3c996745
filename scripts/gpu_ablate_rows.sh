#!/bin/bash
# phase ablation of the row-owner kernel (timing only)
run() { FENRIS_HIP_ROWS=1 python bench.py --steps 10 --warmup 2 --cells ${CELLS:-128} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel'], round(d['roofline']['kernel_avg_ms'],3))"; }
run base
FENRIS_HIP_ABLATE=4 run no_stores
FENRIS_HIP_ABLATE=2 run no_phaseC
FENRIS_HIP_ABLATE=1 run no_phaseB
FENRIS_HIP_ABLATE=6 run no_C_no_stores
FENRIS_HIP_ABLATE=7 run skeleton
