mkdir -p gpurun_out; rm -f gpurun_out/q4.log
run() { echo "== $1" >> gpurun_out/q4.log; shift
  env "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('elem/s %.4g kernel_ms %.3f frac %.4f %s' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['kernel']))" >> gpurun_out/q4.log 2>&1
}
B="timeout 300 python bench.py --steps 5 --warmup 2 --cells 128 --no-cpu-baseline"
for nb in 3 4 5 6 7; do run "NB=$nb" FENRIS_HIP_GATHER_NB=$nb $B; done
for nb in 4 5; do run "NB=$nb wgs=3" FENRIS_HIP_GATHER_NB=$nb FENRIS_HIP_PIPE_WGS_PER_CU=3 $B; done
for nb in 6 7 8; do run "poisson NB=$nb" FENRIS_HIP_GATHER_NB=$nb $B --operator poisson; done
cat gpurun_out/q4.log
