#!/bin/bash
# round 3: ablation matrix of the affine kernels (ring form and second form) on the headline; FENRIS_HIP_ABLATE bits: 1 no global
# stores, 2 no products, 4 no record fetches
OUT=gpurun_out/r3c; mkdir -p $OUT
for ring in 1 0; do
for ab in 0 1 2 4 6 3 5 7; do
    FENRIS_HIP_AFFINE_RING=$ring FENRIS_HIP_ABLATE=$ab $EXTRA_ENV timeout 300 python bench.py --config ${CFG:-ns} --no-traffic --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/b.json")); print("${CFG:-ns} ring=$ring ablate=$ab", round(d["ms_per_step"],4), d["roofline"]["kernel"])
except Exception as e: print("ring=$ring ablate=$ab FAILED", e)
PY
done
done 2>&1 | tee $OUT/ablate_${CFG:-ns}.txt
