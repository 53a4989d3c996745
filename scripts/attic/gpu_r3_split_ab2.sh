#!/bin/bash
OUT=gpurun_out/r3f; mkdir -p $OUT
run() { # label env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --config ${CFG:-ns} --no-traffic --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b.json")); print("${CFG:-ns} $label", round(d["ms_per_step"],4), round(d["roofline"]["frac"],4))
except Exception as e: print("$label FAILED", e)
PY
}
for rep in 1 2; do
run "nosplit" FENRIS_HIP_AFFINE_NO_SPLIT=1
run "split125 2streams" A=1
run "split125 mode1 (records all; rows A, B)" FENRIS_HIP_AFFINE_SPLIT_MODE=1
run "split125 mode2 (records A, B; rows A, B; one stream)" FENRIS_HIP_AFFINE_SPLIT_MODE=2
run "split500 mode1" FENRIS_HIP_AFFINE_SPLIT_MODE=1 FENRIS_HIP_AFFINE_SPLIT_PERMILLE=500
done 2>&1 | tee $OUT/split_ab2.txt
