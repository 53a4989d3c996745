#!/bin/bash
# round 4: lane tables of k_affine_rows rearranged against LDS bank conflicts (on / off), fresh processes alternating
CFG=${1:-ns}
for rep in 1 2 3 4; do
  for t in on off; do
    if [ $t = off ]; then export FENRIS_HIP_NO_LANE_TUNING=1; else unset FENRIS_HIP_NO_LANE_TUNING; fi
    python bench.py --config $CFG --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep tuning $t:', round(d['ms_per_step'],4))"
  done
done
