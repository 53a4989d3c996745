#!/bin/bash
# round 5, C4: the two passes on DISJOINT CU sets (FENRIS_HIP_TWO_PASS_GATHER_CUS = CUs of the row gather, hipExtStreamCreateWithCUMask)
mkdir -p gpurun_out/r5_c4
run() {
  local label=$1; shift
  env "$@" python bench.py --config c4 --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'ms_per_step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3))"
}
run serial FENRIS_HIP_TWO_PASS_CHUNKS=0
for g in 32 48 64 80 96; do
  for ch in 8 16 32; do
    run "gather_cus=$g chunks=$ch" FENRIS_HIP_TWO_PASS_CHUNKS=$ch FENRIS_HIP_TWO_PASS_GATHER_CUS=$g
  done
done
