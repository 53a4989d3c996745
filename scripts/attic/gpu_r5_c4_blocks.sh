mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_c4
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_hex27_mfma.py tests/test_quadratic_elements.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30 > $OUT/tests_blocks.txt
tail -12 $OUT/tests_blocks.txt
for v in 0 1 0 1; do
  FENRIS_HIP_HEX27_TILES16=$v timeout 300 python3 bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-secondary --placement-tries 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles16=$v ms_per_step', round(d['ms_per_step'],3))"
done
