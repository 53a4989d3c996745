#!/bin/bash
# pipelined kernel against the row-owner kernel (FENRIS_HIP_ROWS=1) on the headline workload
run() { python bench.py --steps 10 --warmup 2 --cells ${CELLS:-128} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel'], round(d['roofline']['kernel_avg_ms'],3), 'pattern_s', round(d['config']['pattern_build_s'],3))"; }
run pipelined
FENRIS_HIP_ROWS=1 run rows
run pipelined
FENRIS_HIP_ROWS=1 run rows
CELLS=216 run pipelined216
CELLS=216 FENRIS_HIP_ROWS=1 run rows216
