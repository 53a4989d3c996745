#!/bin/bash
# C3 (Tet4) variants of the pipelined kernel
r() { BENCH_GATHER_ONLY=1 python scripts/bench_configs.py C3 2>/dev/null | head -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', round(d['modes']['gather']['kernel_ms'],3))"; }
r base
FENRIS_HIP_NO_ALIGN=1 r noalign
FENRIS_HIP_ABLATE=16 r nooverlap
FENRIS_HIP_PIPE_WGS_PER_CU=3 r wgs3
FENRIS_HIP_PIPE_WGS_PER_CU=2 r wgs2
FENRIS_HIP_PIPE_JT=1 r jt1
FENRIS_HIP_PIPE_JT=4 r jt4
FENRIS_HIP_GATHER_NB=5 r nb5
FENRIS_HIP_GATHER_MB=96 r mb96
