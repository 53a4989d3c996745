#!/bin/bash
# round 5, C4 first pass: builds of k_hex27_dense_mfma side by side on one box (FENRIS_HIP_LIB), two and three workgroups per CU, two rounds
run() {
  local label=$1 lib=$2 w=$3
  if [ -n "$lib" ]; then export FENRIS_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset FENRIS_HIP_LIB; fi
  FENRIS_HIP_HEX27_WGS_PER_CU=$w python bench.py --config c4 --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label wgs_per_cu=$w ms_per_step', round(d['ms_per_step'],3))"
}
for rep in 1 2; do
  run "round-4 library          " scripts/bin/lib_r4/libfenris_hip.so 2
  run "v0 diet, one round, lb2  " scripts/bin/lib_c4/v0_single_round_lb2_prefetch.so 2
  run "v1 one round, lb3, spills" scripts/bin/lib_c4/v1_single_round_lb3.so 3
  run "v2 two rounds, lb3       " scripts/bin/lib_c4/v2_two_rounds_noprio.so 3
  run "v2 two rounds, lb3       " scripts/bin/lib_c4/v2_two_rounds_noprio.so 2
  run "tree (v2 + setprio)      " "" 3
done
