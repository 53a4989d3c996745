mkdir -p gpurun_out; rm -f gpurun_out/q3.log
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3) > gpurun_out/tests.log 2>&1
run() { echo "== $1" >> gpurun_out/q3.log; shift
  env "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('elem/s %.4g kernel_ms %.3f frac %.4f %s' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['kernel']))" >> gpurun_out/q3.log 2>&1
}
B="timeout 300 python bench.py --steps 5 --warmup 2 --cells 128 --no-cpu-baseline"
run "default" A=1 $B
run "NB=6" FENRIS_HIP_GATHER_NB=6 $B
run "NB=8 QC=4" FENRIS_HIP_GATHER_NB=8 FENRIS_HIP_PIPE_QC=4 $B
run "nosweep" FENRIS_HIP_NO_SWEEP=1 $B
run "NB=8" FENRIS_HIP_GATHER_NB=8 $B
run "NB=10 MB=256" FENRIS_HIP_GATHER_NB=10 FENRIS_HIP_GATHER_MB=256 $B
run "poisson" A=1 $B --operator poisson
for ab in 1 2 4; do run "ablate=$ab" FENRIS_HIP_ABLATE=$ab $B; done
run "poisson nosweep" FENRIS_HIP_NO_SWEEP=1 $B --operator poisson
export FENRIS_HIP_VERBOSE=1; $B 2>&1 | grep "sweep order" | head -2 >> gpurun_out/q3.log
cat gpurun_out/tests.log gpurun_out/q3.log
