run() { BENCH_GATHER_ONLY=1 python scripts/bench_configs.py C3 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['modes']['gather']['kernel_ms'],3))"; }
run production
FENRIS_HIP_DBG_KERNEL=1 run dbg
FENRIS_HIP_ABLATE=512 run verts_from_64_nodes
FENRIS_HIP_ABLATE=4 run no_finalize
FENRIS_HIP_ABLATE=2 run no_phaseC
FENRIS_HIP_ABLATE=1 run no_phaseB
FENRIS_HIP_ABLATE=8 run no_writeout
FENRIS_HIP_ABLATE=519 run no_B_C_F_verts
FENRIS_HIP_TRACE=1 python scripts/bench_configs.py C3 2>&1 | grep "trace\] wave 0" | head -8
