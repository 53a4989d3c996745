# what the driver sees: the default line as the FIRST process on a fresh box, then a second one (gpurun_out/r3q)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3q
mkdir -p $OUT
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_first.json 2> $OUT/bench_first.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_second.json 2> $OUT/bench_second.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3q/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0])
    r=d['roofline']
    print(os.path.basename(f), round(d['ms_per_step'],3), round(r['frac'],3), r['kernel_min_ms'], r['traffic'], d['config']['device_settle'], d['config']['placement_probe'], {k:round(v.get('ms',0),3) for k,v in d.get('secondary',{}).items()})
PY
