#!/bin/bash
r() { BENCH_GATHER_ONLY=1 python scripts/bench_configs.py C4 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', round(d['modes']['gather']['kernel_ms'],2))"; }
r offset
FENRIS_HIP_ABLATE=8 r lockstep
r offset
