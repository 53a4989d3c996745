#!/bin/bash
# round 3: the ring form of the affine kernel against the second form (same box, same process order), tests first
OUT=gpurun_out/r3b; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log; tail -3 $OUT/gputest.log
for rep in 1 2; do
for ring in 1 0; do
  for cfg in ns c2; do
    FENRIS_HIP_AFFINE_RING=$ring timeout 300 python bench.py --config $cfg --no-traffic --no-cpu-baseline > $OUT/bench_${cfg}_ring${ring}_$rep.json 2> $OUT/bench_${cfg}_ring${ring}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${cfg}_ring${ring}_$rep.json")); print("$cfg ring=$ring rep=$rep", round(d["ms_per_step"],4), round(d["roofline"]["frac"],4), d["roofline"]["kernel"])
except Exception as e: print("$cfg ring=$ring FAILED", e)
PY
  done
done
done
./scripts/bin/ubench_mix > $OUT/ubench_mix.jsonl 2>&1
