mkdir -p gpurun_out; rm -f gpurun_out/quick2.log
run() { echo "== $1" >> gpurun_out/quick2.log; shift
  env "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('elem/s %.4g kernel_ms %.3f frac %.4f %s' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['kernel']))" >> gpurun_out/quick2.log 2>&1
}
B="timeout 300 python bench.py --steps 5 --warmup 2 --cells 128 --no-cpu-baseline --operator poisson"
for jt in 2 4; do for nb in 6 8 12 16; do for qc in 4 8; do run "poisson JT=$jt NB=$nb QC=$qc" FENRIS_HIP_PIPE_JT=$jt FENRIS_HIP_GATHER_NB=$nb FENRIS_HIP_PIPE_QC=$qc $B; done; done; done
B="timeout 300 python bench.py --steps 5 --warmup 2 --cells 128 --no-cpu-baseline"
for nb in 5 6 7; do run "elast JT=2 NB=$nb QC=8" FENRIS_HIP_PIPE_JT=2 FENRIS_HIP_GATHER_NB=$nb FENRIS_HIP_PIPE_QC=8 $B; done
run "elast JT=2 NB=6 QC=8 wgs=3" FENRIS_HIP_PIPE_WGS_PER_CU=3 FENRIS_HIP_PIPE_JT=2 FENRIS_HIP_GATHER_NB=6 FENRIS_HIP_PIPE_QC=8 $B
run "elast JT=2 NB=6 QC=8 wgs=1" FENRIS_HIP_PIPE_WGS_PER_CU=1 FENRIS_HIP_PIPE_JT=2 FENRIS_HIP_GATHER_NB=6 FENRIS_HIP_PIPE_QC=8 $B
cat gpurun_out/quick2.log
