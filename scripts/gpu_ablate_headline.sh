#!/bin/bash
# phase ablation of the headline kernel (instrumented twin of the production instantiation; results are wrong, timing only)
run() { python bench.py --steps 10 --warmup 2 --cells ${CELLS:-128} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['roofline']['kernel_avg_ms'],3))"; }
run production
FENRIS_HIP_DBG_KERNEL=1 run dbg_base
FENRIS_HIP_ABLATE=32 run plain_stores
FENRIS_HIP_ABLATE=64 run atomics_conflict_free
FENRIS_HIP_ABLATE=96 run plain_stores_conflict_free
FENRIS_HIP_ABLATE=4 run no_finalize
FENRIS_HIP_ABLATE=2 run no_phaseC
FENRIS_HIP_ABLATE=1 run no_phaseB
FENRIS_HIP_ABLATE=8 run no_writeout
FENRIS_HIP_ABLATE=7 run no_B_C_finalize
FENRIS_HIP_DBG_KERNEL=1 run dbg_base
