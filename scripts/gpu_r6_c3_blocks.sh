#!/bin/bash
# round 6: Tet4 row-owner tables with 13 nodes / 352 entries per position first (then 11 / 288, 9 / 256, 7 / 224) -- tests, fuzz, C3
mkdir -p gpurun_out/r6_c3
F='HIP version\|ROCm version\|Hostname\|Librccl\|RCCL\|amdgpu.ids'
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "not bench_launch" 2>&1 | grep -v "$F" | tail -3
timeout 900 python3 scripts/fuzz_gather.py 6000 3000000 big 2>&1 | tail -1
for r in 1 2 3; do timeout 200 python3 bench.py --config c3 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['config'].get('first_assembly_s'))"; done
python3 scripts/check_full_size.py 2>&1 | grep "c3"
