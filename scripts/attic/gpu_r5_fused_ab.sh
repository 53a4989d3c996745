#!/bin/bash
# round 5: the fused records wave of k_affine_rows (FENRIS_HIP_AFFINE_FUSED=1: vertex tables at the pattern build, a seventh wavefront forms the
# element records by LDS-DMA) against the separate records kernel, inside one context (scripts/ab_in_context.py), with the records wave's
# DMA levels / arithmetic switched off (FENRIS_HIP_AFFINE_REC_ABLATE: 1 no DMA, 2 no arithmetic, 4 no table rows, 8 no vertex gathers -- timing
# only), and the per-role cycle trace of the instrumented instantiation.  Result: profiles/r05_fused_records_experiment.txt
mkdir -p gpurun_out/r5_fused
export FENRIS_HIP_AFFINE_FUSED=1
for cfg in ns c2; do
  python scripts/ab_in_context.py --config $cfg "fused:" "separate:FENRIS_HIP_AFFINE_FUSED=0" "fused_nodma:FENRIS_HIP_AFFINE_REC_ABLATE=1" "fused_nomath:FENRIS_HIP_AFFINE_REC_ABLATE=2" \
      "fused_neither:FENRIS_HIP_AFFINE_REC_ABLATE=3" "fused_no_table_rows:FENRIS_HIP_AFFINE_REC_ABLATE=4" "fused_no_vertex_gathers:FENRIS_HIP_AFFINE_REC_ABLATE=8" 2>&1 | grep variant | sed "s/^/$cfg /" | tee gpurun_out/r5_fused/ab_$cfg.txt
  for f in 1 0; do
    echo "--- per-role trace, $cfg, FENRIS_HIP_AFFINE_FUSED=$f (wave 0 row wave, 1 loader, 2 store wave: work between barriers | - | at the barrier; wave 3 = records wave: landed-wait | arithmetic | DMA issue | barrier)"
    FENRIS_HIP_AFFINE_FUSED=$f FENRIS_HIP_TRACE=1 FENRIS_HIP_ABLATE=16 python bench.py --config $cfg --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle --steps 5 2>&1 | grep -i "trace" | grep -v " 0 cycles" | cut -c1-160 | tail -12
  done
done
