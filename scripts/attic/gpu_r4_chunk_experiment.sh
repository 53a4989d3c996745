#!/bin/bash
# round 4, review item 4: do the "levels" of the headline (4.6 ... 5.4 ms, depending on the physical memory behind the values array) come
# from the 768 contiguous write fronts ~25 MB apart?  Positions dealt to the workgroups in chunks of C (FENRIS_HIP_AFFINE_CHUNK) put all
# concurrent stores into one moving window instead.
#  (1) correctness of the chunked form, (2) in one context on the same buffers, (3) first placements of fresh processes (no settle, no probe)
mkdir -p gpurun_out/r4
echo "== correctness (chunk 16, 50)"
for c in 16 50; do FENRIS_HIP_AFFINE_CHUNK=$c python -m pytest tests/test_affine.py tests/test_full_size_slabs.py -m gpu -q -k "not bench" 2>&1 | tail -1; done
FENRIS_HIP_AFFINE_CHUNK=32 python scripts/check_full_size.py ns c2 2>&1 | tail -2
echo "== in one context"
python scripts/ab_in_context.py --config ns "contiguous:" "chunk8:FENRIS_HIP_AFFINE_CHUNK=8" "chunk16:FENRIS_HIP_AFFINE_CHUNK=16" "chunk32:FENRIS_HIP_AFFINE_CHUNK=32" "chunk64:FENRIS_HIP_AFFINE_CHUNK=64" "chunk128:FENRIS_HIP_AFFINE_CHUNK=128" "chunk31:FENRIS_HIP_AFFINE_CHUNK=31" 2>&1 | tail -8
echo "== fresh processes, first placement (--placement-tries 0 --no-settle)"
for rep in 1 2 3 4 5; do
  for c in 0 32; do
    FENRIS_HIP_AFFINE_CHUNK=$c python bench.py --config ns --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep chunk $c:', round(d['ms_per_step'],3), 'ms')"
  done
done
