run() { BENCH_GATHER_ONLY=1 python scripts/bench_configs.py C3 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); m=d['modes']['gather']; print('$1', m['kernel'], round(m['kernel_ms'],3))"; }
run default
FENRIS_HIP_GATHER_MB=192 FENRIS_HIP_PIPE_JT=4 FENRIS_HIP_GATHER_NB=8 run mb192_nb8
FENRIS_HIP_GATHER_MB=256 FENRIS_HIP_PIPE_JT=4 FENRIS_HIP_GATHER_NB=8 run mb256_nb8
FENRIS_HIP_GATHER_MB=96 run mb96
FENRIS_HIP_GATHER_MB=160 FENRIS_HIP_PIPE_JT=4 FENRIS_HIP_GATHER_NB=7 run mb160_nb7
FENRIS_HIP_PIPE_WGS_PER_CU=4 run wgs4
