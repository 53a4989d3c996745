# A/B and ablation timings of the affine-element kernel (k_affine_rows) on the headline configuration.
# usage (GPU box, repo root): bash scripts/gpu_affine_ab.sh [cells]
CELLS=${1:-216}
OUT=gpurun_out/affine_ab.txt
mkdir -p gpurun_out; : > $OUT
run() {  # label, env assignments...
  label=$1; shift
  line=$(env "$@" python bench.py --steps 10 --warmup 2 --cells $CELLS --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel'], '%.3f ms avg, %.3f min, frac %.3f' % (r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))")
  echo "$label: $line" | tee -a $OUT
}
run "v2 default"
run "v1 (k_gather_affine)" FENRIS_HIP_AFFINE_V1=1
run "v2 wgs/cu=2" FENRIS_HIP_AFFINE_WGS_PER_CU=2
run "v2 wgs/cu=4" FENRIS_HIP_AFFINE_WGS_PER_CU=4
run "v2 ablate 1 (no global stores)" FENRIS_HIP_ABLATE=1
run "v2 ablate 2 (no sandwich)" FENRIS_HIP_ABLATE=2
run "v2 ablate 4 (no slot records)" FENRIS_HIP_ABLATE=4
run "v2 ablate 8 (no lane reload)" FENRIS_HIP_ABLATE=8
run "v2 ablate 16 (dbg instantiation only)" FENRIS_HIP_ABLATE=16
run "v2 ablate 3" FENRIS_HIP_ABLATE=3
run "v2 ablate 14" FENRIS_HIP_ABLATE=14
run "v2 ablate 15" FENRIS_HIP_ABLATE=15
run "v2 staged store wave (64)" FENRIS_HIP_ABLATE=64
run "v2 staged, no stores (65)" FENRIS_HIP_ABLATE=65
