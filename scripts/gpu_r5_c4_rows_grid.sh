for g in 256 512 768 1024 2048 131072; do
  FENRIS_HIP_TWO_PASS_ROWS_GRID=$g python bench.py --config c4 --no-traffic --no-cpu-baseline --no-secondary --placement-tries 0 --no-settle --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial rows_grid=$g ms_per_step', round(d['ms_per_step'],3))"
done
