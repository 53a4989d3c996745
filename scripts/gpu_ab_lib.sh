#!/bin/bash
# two builds of libfenris_hip.so side by side on one box, fresh processes, alternating: first placements (no settle, no probe) and probed
#   scripts/gpu_ab_lib.sh <other-lib.so> [config]
OTHER=$1; CFG=${2:-ns}
for rep in 1 2 3; do
  for which in tree other; do
    if [ $which = other ]; then export FENRIS_HIP_LIB=$GRAFT_REPO_ROOT/$OTHER; else unset FENRIS_HIP_LIB; fi
    python bench.py --config $CFG --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep $which first placement:', round(d['ms_per_step'],3))"
  done
done
for which in tree other; do
  if [ $which = other ]; then export FENRIS_HIP_LIB=$GRAFT_REPO_ROOT/$OTHER; else unset FENRIS_HIP_LIB; fi
  python bench.py --config $CFG --no-traffic --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which default line:', round(d['ms_per_step'],3), d['config']['placement_probe']['values_ms_seen'])"
done
