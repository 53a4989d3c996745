#!/bin/bash
# quick A/B of the pipelined kernel variants at 128^3
for ab in 0 16; do
  echo "== ablate=$ab"
  FENRIS_HIP_ABLATE=$ab python bench.py --steps 10 --warmup 2 --cells 128 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'])"
done
FENRIS_HIP_TRACE=1 python bench.py --steps 5 --warmup 2 --cells 128 --no-cpu-baseline 2>&1 | grep trace
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
