#!/bin/bash
# round 4: ablations of k_hex8_rows on ns-perturbed (bits: 1 no global stores, 2 no phase C products, 4 no phase B)
mkdir -p gpurun_out/r4
for ab in ${AB_LIST:-0 1 2 4 6 7} ${EXTRA_AB}; do
  FENRIS_HIP_ABLATE=$ab $EXTRA_ENV timeout 300 python bench.py --config ns-perturbed --no-traffic --no-cpu-baseline --placement-tries 0 --no-settle --steps 10 > gpurun_out/r4/ab_$ab.json 2> gpurun_out/r4/ab_$ab.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r4/ab_$ab.json").read().strip().splitlines()[-1])
    print("ablate $ab:", round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"])
except Exception as e:
    print("ablate $ab: failed", e)
PY
done
