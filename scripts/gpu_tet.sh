mkdir -p gpurun_out; rm -f gpurun_out/tet.log
run() { echo "== $1" >> gpurun_out/tet.log; shift
  env "$@" python scripts/bench_configs.py C3 2>>gpurun_out/tet.err | python -c "
import sys,json
for line in sys.stdin:
    d=json.loads(line); g=d['modes']['gather']; print(d['config'][:12], 'gather ms %.3f' % g['kernel_ms'], g['kernel'])" >> gpurun_out/tet.log 2>&1
}
export FENRIS_HIP_VERBOSE=1
run "default" A=1
run "QC=1" FENRIS_HIP_PIPE_QC=1
run "QC=1 NB=4" FENRIS_HIP_PIPE_QC=1 FENRIS_HIP_GATHER_NB=4
run "QC=1 NB=16 MB=256" FENRIS_HIP_PIPE_QC=1 FENRIS_HIP_GATHER_NB=16 FENRIS_HIP_GATHER_MB=256
run "QC=1 wgs=2" FENRIS_HIP_PIPE_QC=1 FENRIS_HIP_PIPE_WGS_PER_CU=2
run "QC=1 wgs=4" FENRIS_HIP_PIPE_QC=1 FENRIS_HIP_PIPE_WGS_PER_CU=4
run "nopipe" FENRIS_HIP_NO_PIPE=1
for ab in 1 2 4 8 15; do run "QC=1 ablate=$ab" FENRIS_HIP_PIPE_QC=1 FENRIS_HIP_ABLATE=$ab; done
grep "fenris_hip" gpurun_out/tet.err | sort | uniq -c | head -20
cat gpurun_out/tet.log
