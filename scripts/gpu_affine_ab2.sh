CELLS=${1:-216}
OUT=gpurun_out/affine_ab2.txt
mkdir -p gpurun_out; : > $OUT
run() {
  label=$1; shift
  line=$(env "$@" python bench.py --steps 10 --warmup 2 --cells $CELLS --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel'], '%.3f ms avg, %.3f min, frac %.3f' % (r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))")
  echo "$label: $line" | tee -a $OUT
}
run "default"
run "16 dbg" FENRIS_HIP_ABLATE=16
run "32 J arithmetic off, loads on" FENRIS_HIP_ABLATE=32
run "128 J vertices from cache" FENRIS_HIP_ABLATE=128
run "160 both" FENRIS_HIP_ABLATE=160
run "4 no J at all" FENRIS_HIP_ABLATE=4
run "1 no stores" FENRIS_HIP_ABLATE=1
run "33 no stores, J arithmetic off" FENRIS_HIP_ABLATE=33
run "129 no stores, J vertices from cache" FENRIS_HIP_ABLATE=129
run "5 no stores, no J" FENRIS_HIP_ABLATE=5
