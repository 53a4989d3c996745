# usage: bash scripts/gpu_sweep.sh   (on the GPU box, from the repo root)
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -30) > gpurun_out/tests.log 2>&1
rm -f gpurun_out/sweep.log
run() { # label, env..., -- bench args
  echo "== $1" >> gpurun_out/sweep.log; shift
  env "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('elem/s %.4g  kernel_ms %.3f  step_ms %.3f  frac %.4f' % (d['value'], d['roofline']['kernel_avg_ms'], d['ms_per_step'], d['roofline']['frac']))" >> gpurun_out/sweep.log 2>&1
}
B="timeout 300 python bench.py --steps 5 --warmup 2 --cells 128 --no-cpu-baseline"
run "gather NB=8 LDS=78"  FENRIS_HIP_GATHER_NB=8 FENRIS_HIP_GATHER_LDS_KB=78 $B
run "gather NB=4 LDS=40"  FENRIS_HIP_GATHER_NB=4 FENRIS_HIP_GATHER_LDS_KB=40 $B
run "gather NB=4 LDS=52"  FENRIS_HIP_GATHER_NB=4 FENRIS_HIP_GATHER_LDS_KB=52 $B
run "gather NB=2 LDS=30"  FENRIS_HIP_GATHER_NB=2 FENRIS_HIP_GATHER_LDS_KB=30 $B
run "gather NB=1 LDS=20"  FENRIS_HIP_GATHER_NB=1 FENRIS_HIP_GATHER_LDS_KB=20 $B
run "gather NB=16 LDS=150" FENRIS_HIP_GATHER_NB=16 FENRIS_HIP_GATHER_LDS_KB=150 $B
run "gather NB=8 LDS=78 nopipe"  FENRIS_HIP_NO_PIPE=1 FENRIS_HIP_GATHER_NB=8 FENRIS_HIP_GATHER_LDS_KB=78 $B
run "gather NB=8 LDS=78 wgs=1"  FENRIS_HIP_PIPE_WGS_PER_CU=1 FENRIS_HIP_GATHER_NB=8 FENRIS_HIP_GATHER_LDS_KB=78 $B
run "gather NB=12 LDS=100"  FENRIS_HIP_GATHER_NB=12 FENRIS_HIP_GATHER_LDS_KB=100 $B
run "atomic" A=1 $B --scatter atomic
run "colored" A=1 $B --scatter colored
run "poisson gather" A=1 $B --operator poisson
cat gpurun_out/tests.log gpurun_out/sweep.log
