#!/bin/bash
# A/B of pipelined-kernel instantiations on one box: production (compile-time point count), generic, instrumented
run() { python bench.py --steps 10 --warmup 2 --cells ${CELLS:-128} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['roofline']['kernel_avg_ms'],3))"; }
run fullq
FENRIS_HIP_NO_FULLQ=1 run generic
FENRIS_HIP_DBG_KERNEL=1 run dbg
run fullq
FENRIS_HIP_NO_FULLQ=1 run generic
CELLS=216 run fullq216
FENRIS_HIP_NO_FULLQ=1 CELLS=216 run generic216
