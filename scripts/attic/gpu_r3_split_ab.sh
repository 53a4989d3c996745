#!/bin/bash
OUT=gpurun_out/r3f; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log; tail -3 $OUT/gputest.log
run() { # label env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --config ${CFG:-ns} --no-traffic --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b.json")); print("${CFG:-ns} $label", round(d["ms_per_step"],4), round(d["roofline"]["frac"],4), "pattern_s", round(d["config"]["pattern_build_s"],3))
except Exception as e: print("$label FAILED", e)
PY
}
for rep in 1 2; do
run "nosplit rows" FENRIS_HIP_AFFINE_NO_SPLIT=1
run "split125 rows" A=1
run "split60 rows" FENRIS_HIP_AFFINE_SPLIT_PERMILLE=60
run "split250 rows" FENRIS_HIP_AFFINE_SPLIT_PERMILLE=250
run "split125 ring" FENRIS_HIP_AFFINE_RING=1
run "nosplit ring" FENRIS_HIP_AFFINE_RING=1 FENRIS_HIP_AFFINE_NO_SPLIT=1
done 2>&1 | tee $OUT/split_ab.txt
CFG=c5 run "c5 split125 rows" A=1
CFG=c5 run "c5 nosplit rows" FENRIS_HIP_AFFINE_NO_SPLIT=1
CFG=c2 run "c2 split125 rows" A=1
CFG=c2 run "c2 nosplit rows" FENRIS_HIP_AFFINE_NO_SPLIT=1
