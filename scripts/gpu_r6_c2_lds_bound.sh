#!/bin/bash
# round 6: C2 -- what fewer LDS record reads could buy at most, in the PRODUCTION instantiation (timing only, wrong sums): FENRIS_HIP_C2_EXP 1 = a lane
# reads ONE element record (three 16-byte reads) instead of two, 2 = none at all (operands from registers)
mkdir -p gpurun_out/r6_c2
timeout 600 python3 scripts/ab_in_context.py --config c2 --rounds 5 --reps 20 "prod:" "one_record:FENRIS_HIP_C2_EXP=1" "no_record:FENRIS_HIP_C2_EXP=2" 2>&1 | grep variant | tee gpurun_out/r6_c2/lds_bound.txt
