export TMPDIR=/tmp
timeout 900 python3 bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ns', round(d['ms_per_step'],3), round(d['roofline']['frac'],3))
for k,v in d['secondary'].items(): print(k, round(v['ms'],3), round(v['frac'],3), v['placement_probe']['values_ms_seen'], v.get('device_settle'), round(v['seconds_total'],1))"
