#!/bin/bash
# round 3: one environment variable swept over values on one configuration: VAR, VALUES (space separated), CFG, EXTRA_ENV
OUT=gpurun_out/r3d; mkdir -p $OUT
for v in $VALUES; do
    env $EXTRA_ENV $VAR=$v timeout 300 python bench.py --config ${CFG:-ns} --no-traffic --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/b.json")); print("${CFG:-ns} $EXTRA_ENV $VAR=$v", round(d["ms_per_step"],4), round(d["roofline"]["frac"],4))
except Exception as e: print("$VAR=$v FAILED", e)
PY
done 2>&1 | tee -a $OUT/sweep_${VAR}.txt
