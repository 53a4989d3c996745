# three default bench.py lines back to back (PMC passes after the timed region): gpurun_out/r3y/bench_{1,2,3}.json
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3y
mkdir -p $OUT
for i in 1 2 3; do
  python bench.py --no-cpu-baseline > $OUT/bench_$i.json 2> $OUT/bench_$i.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3y/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0])
    r=d['roofline']
    print(os.path.basename(f), round(d['ms_per_step'],3), round(r['frac'],3), r['kernel_min_ms'], r['traffic'], d['config']['placement_probe'], {k:round(v.get('ms',0),3) for k,v in d.get('secondary',{}).items()})
PY
