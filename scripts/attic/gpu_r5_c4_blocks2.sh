mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_c4
export TMPDIR=/tmp
for v in 0 1; do
  echo "=== tiles16=$v"
  FENRIS_HIP_HEX27_TILES16=$v FENRIS_HIP_TRACE=1 timeout 300 python3 bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-secondary --placement-tries 0 --no-settle 2>&1 | grep -i "trace" | cut -c1-200 | tail -14
done
