mkdir -p gpurun_out; rm -f gpurun_out/sweep2.log gpurun_out/sweep2.err
run() { echo "== $1" >> gpurun_out/sweep2.log; shift
  env "$@" 2>>gpurun_out/sweep2.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('elem/s %.4g  kernel_ms %.3f  frac %.4f  %s' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['kernel']))" >> gpurun_out/sweep2.log 2>&1
}
export FENRIS_HIP_VERBOSE=1
export FENRIS_HIP_GATHER_LDS_KB=150
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/tests.log 2>&1
B="timeout 300 python bench.py --steps 5 --warmup 2 --cells 128 --no-cpu-baseline"
for nb in 4 6 8; do for qc in 1 2 4 8; do
  run "pipe NB=$nb QC=$qc" FENRIS_HIP_GATHER_NB=$nb FENRIS_HIP_PIPE_QC=$qc $B
done; done
run "poisson NB=8 QC=2" FENRIS_HIP_GATHER_NB=8 FENRIS_HIP_PIPE_QC=2 $B --operator poisson
run "poisson NB=8 QC=8" FENRIS_HIP_GATHER_NB=8 FENRIS_HIP_PIPE_QC=8 $B --operator poisson
grep "pipelined" gpurun_out/sweep2.err | sort | uniq -c; cat gpurun_out/tests.log gpurun_out/sweep2.log
