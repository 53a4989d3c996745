#!/bin/bash
# quick sweep of the pipelined kernel at 128^3: JT x NB
run() { python bench.py --steps 10 --warmup 2 --cells 128 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
run base
FENRIS_HIP_PIPE_JT=4 run jt4
FENRIS_HIP_GATHER_NB=6 run nb6
FENRIS_HIP_GATHER_NB=6 FENRIS_HIP_PIPE_JT=4 run nb6_jt4
FENRIS_HIP_GATHER_NB=5 run nb5
FENRIS_HIP_ABLATE=16 run noverlap
run base
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
