#!/bin/bash
# A/B: production pipelined kernel vs the instrumented (DBG) instantiation with all hooks idle
run() { python bench.py --steps 10 --warmup 2 --cells ${CELLS:-128} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['roofline']['kernel_avg_ms'],3))"; }
run prod
FENRIS_HIP_DBG_KERNEL=1 run dbg
run prod
FENRIS_HIP_DBG_KERNEL=1 run dbg
CELLS=216 run prod216
FENRIS_HIP_DBG_KERNEL=1 CELLS=216 run dbg216
FENRIS_HIP_TRACE=1 python bench.py --steps 3 --warmup 1 --cells 128 --no-cpu-baseline 2>&1 | grep "wave 0" | head -3
