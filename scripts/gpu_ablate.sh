mkdir -p gpurun_out; rm -f gpurun_out/ablate.log
run() { echo "== $1" >> gpurun_out/ablate.log; shift
  env "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms %.3f  %s' % (d['roofline']['kernel_avg_ms'], d['roofline']['kernel']))" >> gpurun_out/ablate.log 2>&1
}
B="timeout 300 python bench.py --steps 5 --warmup 2 --cells 128 --no-cpu-baseline"
for ab in 0 1 2 4 8 3 6 7 14 15 12; do run "ablate=$ab" FENRIS_HIP_ABLATE=$ab $B; done
for ab in 0 1 2 7 15; do run "poisson ablate=$ab" FENRIS_HIP_ABLATE=$ab $B --operator poisson; done
cat gpurun_out/ablate.log
