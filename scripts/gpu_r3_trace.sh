#!/bin/bash
# round 3: per-role cycle counters of k_affine_ring on the headline, for a few ablations
OUT=gpurun_out/r3e; mkdir -p $OUT
for ab in ${ABLATES:-0 2 4 6}; do
    echo "== ablate=$ab $EXTRA_ENV"
    env $EXTRA_ENV FENRIS_HIP_TRACE=1 FENRIS_HIP_ABLATE=$ab timeout 300 python bench.py --config ${CFG:-ns} --no-traffic --no-cpu-baseline --steps 5 --warmup 1 2> $OUT/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', round(d['ms_per_step'],4))"
    grep "trace" $OUT/b.err
done 2>&1 | tee -a $OUT/trace_${CFG:-ns}.txt
