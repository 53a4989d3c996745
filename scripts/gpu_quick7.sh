#!/bin/bash
run() { python bench.py --steps 10 --warmup 2 --cells ${CELLS:-128} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
run aligned
FENRIS_HIP_NO_ALIGN=1 run unaligned
FENRIS_HIP_GATHER_NB=6 run aligned_nb6
run aligned
CELLS=216 run aligned216
FENRIS_HIP_NO_ALIGN=1 CELLS=216 run unaligned216
FENRIS_HIP_VERBOSE=1 python bench.py --steps 2 --warmup 1 --cells 128 --no-cpu-baseline 2>&1 | grep fenris_hip | head
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
