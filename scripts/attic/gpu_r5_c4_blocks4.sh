mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_c4
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_hex27_mfma.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12
for v in 1 0 1 0; do
  FENRIS_HIP_HEX27_BLOCKS=$v timeout 300 python3 bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-secondary --placement-tries 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks=$v ms_per_step', round(d['ms_per_step'],3))"
done
FENRIS_HIP_HEX27_BLOCKS=1 FENRIS_HIP_TRACE=1 timeout 300 python3 bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-secondary --placement-tries 0 --no-settle 2>&1 | grep -i "trace" | cut -c1-200 | tail -10
