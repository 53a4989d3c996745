#!/bin/bash
# round 5, C4 (Hex27 NeoHookean): the overlapped two-pass assembly (FENRIS_HIP_TWO_PASS_CHUNKS) against the serial form, fresh processes
mkdir -p gpurun_out/r5_c4
run() {  # label, env...
  local label=$1; shift
  env "$@" python bench.py --config c4 --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'ms_per_step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3))"
}
run serial FENRIS_HIP_TWO_PASS_CHUNKS=0
for ch in 4 8 16 32 64; do
  for th in 256 64; do
    run "chunks=$ch gather_threads=$th" FENRIS_HIP_TWO_PASS_CHUNKS=$ch FENRIS_HIP_TWO_PASS_GATHER_THREADS=$th
  done
done
run serial_again FENRIS_HIP_TWO_PASS_CHUNKS=0
