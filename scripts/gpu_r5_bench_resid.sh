export TMPDIR=/tmp
for v in 0 1; do
FENRIS_HIP_NO_MOMENT_RESIDUAL=$v timeout 900 python3 bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sv=d['secondary_vectors']
print('NO_MOMENT=$v residual_ms', round(sv['residual_ms'],4), 'energy', round(sv['energy_ms_blocking_call'],4), 'ns', round(d['ms_per_step'],3))"
done
