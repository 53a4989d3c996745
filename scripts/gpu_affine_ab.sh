# A/B and ablation timings of the affine-element kernel (k_affine_rows) on the headline configuration.
# usage (GPU box, repo root): bash scripts/gpu_affine_ab.sh [cells] [config]   (config: ns (default) or c2)
# FENRIS_HIP_ABLATE bits (instrumented instantiation, wrong results): 1 no global stores, 2 no sandwich products, 4 no record
# fetches, 16 nothing switched off, 128 every record from the first 4096 (no HBM reads, same instructions)
CELLS=${1:-216}
CFG=${2:-ns}
OUT=gpurun_out/affine_ab_$CFG.txt
mkdir -p gpurun_out; : > $OUT
run() {  # label, env assignments...
  label=$1; shift
  line=$(env "$@" timeout 120 python bench.py --config $CFG --steps 10 --warmup 2 --cells $CELLS --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel'], '%.3f ms avg, %.3f min, frac %.3f' % (r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))")
  echo "$label: $line" | tee -a $OUT
}
run "default"
run "default again"
run "wgs/cu=2" FENRIS_HIP_AFFINE_WGS_PER_CU=2
run "wgs/cu=4" FENRIS_HIP_AFFINE_WGS_PER_CU=4
run "ablate 16 (dbg instantiation only)" FENRIS_HIP_ABLATE=16
run "ablate 1 (no global stores)" FENRIS_HIP_ABLATE=1
run "ablate 2 (no sandwich)" FENRIS_HIP_ABLATE=2
run "ablate 4 (no record fetches)" FENRIS_HIP_ABLATE=4
run "ablate 5 (no stores, no records)" FENRIS_HIP_ABLATE=5
run "ablate 6 (stores, staging and barriers only)" FENRIS_HIP_ABLATE=6
run "ablate 7 (skeleton)" FENRIS_HIP_ABLATE=7
run "no lane dedupe" FENRIS_HIP_NO_LANE_DEDUPE=1
