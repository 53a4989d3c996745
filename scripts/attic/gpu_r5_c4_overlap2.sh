#!/bin/bash
# round 5, C4: overlapped two-pass assembly with a SMALL persistent grid for the row gather (it must not flood the CUs: the first pass needs
# its two workgroups per CU), chunks x gather grid x gather workgroup size
mkdir -p gpurun_out/r5_c4
run() {
  local label=$1; shift
  env "$@" python bench.py --config c4 --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'ms_per_step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3))"
}
run serial FENRIS_HIP_TWO_PASS_CHUNKS=0
for ch in 8 16 32; do
  for cfg in "256 256" "256 512" "256 1024" "64 1024" "64 2048" "128 1024"; do
    set -- $cfg
    run "chunks=$ch gather_threads=$1 gather_grid=$2" FENRIS_HIP_TWO_PASS_CHUNKS=$ch FENRIS_HIP_TWO_PASS_GATHER_THREADS=$1 FENRIS_HIP_TWO_PASS_ROWS_GRID=$2
  done
done
