#!/bin/bash
# round 5: wave priorities in k_affine_rows (store wave / loader wave raised with s_setprio), builds side by side on one box (FENRIS_HIP_LIB):
# first placements (no settle, no probe) of ns, then c2
CFG=${1:-ns}
run() {
  local label=$1 lib=$2
  if [ -n "$lib" ]; then export FENRIS_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset FENRIS_HIP_LIB; fi
  python bench.py --config $CFG --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG $label first placement ms_per_step', round(d['ms_per_step'],4))"
}
for rep in 1 2 3; do
  run "tree (no priorities) " ""
  run "store 3              " scripts/bin/lib_prio/store3_loader0.so
  run "store 3, loader 2    " scripts/bin/lib_prio/store3_loader2.so
  run "store 2              " scripts/bin/lib_prio/store2_loader0.so
  run "store 1              " scripts/bin/lib_prio/store1_loader0.so
done
