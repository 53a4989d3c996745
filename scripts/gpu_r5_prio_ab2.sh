#!/bin/bash
CFG=${1:-ns}
run() {
  local label=$1 lib=$2
  if [ -n "$lib" ]; then export FENRIS_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset FENRIS_HIP_LIB; fi
  python bench.py --config $CFG --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG $label first placement ms_per_step', round(d['ms_per_step'],4))"
}
for rep in 1 2; do
  run "tree (no priorities) " ""
  for v in store3_loader2 store3_loader3 store2_loader3 store3_loader1 store1_loader1 store0_loader2; do run "$v" scripts/bin/lib_prio/$v.so; done
done
