mkdir -p gpurun_out; rm -f gpurun_out/quick.log
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5) > gpurun_out/tests.log 2>&1
run() { echo "== $1" >> gpurun_out/quick.log; shift
  env "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('elem/s %.4g kernel_ms %.3f frac %.4f %s' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['kernel']))" >> gpurun_out/quick.log 2>&1
}
B="timeout 300 python bench.py --steps 5 --warmup 2 --cells 128 --no-cpu-baseline"
run "default" A=1 $B
for jt in 1 2 4; do for nb in 6 8 12; do for qc in 4 8; do run "JT=$jt NB=$nb QC=$qc" FENRIS_HIP_PIPE_JT=$jt FENRIS_HIP_GATHER_NB=$nb FENRIS_HIP_PIPE_QC=$qc $B; done; done; done
run "poisson default" A=1 $B --operator poisson
cat gpurun_out/tests.log gpurun_out/quick.log
