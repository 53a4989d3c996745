export TMPDIR=/tmp
timeout 600 python3 bench.py --config c5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('standalone c5', round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['config']['placement_probe']['values_ms_seen'])"
timeout 900 python3 bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ns', round(d['ms_per_step'],3), round(d['roofline']['frac'],3))
v=d['secondary']['c5']; print('secondary c5', round(v['ms'],3), round(v['frac'],3), v['placement_probe']['values_ms_seen'])"
timeout 600 python3 bench.py --config c5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('standalone c5', round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['config']['placement_probe']['values_ms_seen'])"
