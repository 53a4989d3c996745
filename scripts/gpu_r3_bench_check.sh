#!/bin/bash
# round 3: the default bench line (traffic children, secondary configurations, CPU baseline), the N = 2 self-launch on one device
# (validation mode) and its refusal without it
OUT=gpurun_out/r3h; mkdir -p $OUT
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"; tail -3 $OUT/bench_default.err
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print("ns", round(d["ms_per_step"],4), round(d["roofline"]["frac"],4), "traffic", d["roofline"]["traffic"], "pattern_s", d["config"]["pattern_build_s"], "warm_s", d["config"]["module_warmup_s"])
for k,v in d.get("secondary",{}).items(): print(k, {x: (round(y,4) if isinstance(y,float) else y) for x,y in v.items() if x in ("ms","frac","kernel","error","seconds_total")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
python bench.py --gpus 2 --steps 3 --warmup 1 --cells 24 > $OUT/bench_n2_refused.json 2> $OUT/bench_n2_refused.err; echo "n2 without devices rc=$? (expected 2)"; tail -1 $OUT/bench_n2_refused.err
FENRIS_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --cells 24 > $OUT/bench_n2_share.json 2> $OUT/bench_n2_share.err; echo "n2 share rc=$?"; tail -2 $OUT/bench_n2_share.err; head -c 900 $OUT/bench_n2_share.json; echo
FENRIS_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --config c5 --cells 32 > $OUT/bench_c5_n2_share.json 2> $OUT/bench_c5_n2_share.err; echo "c5 n2 share rc=$?"; python -c "
import json; d=json.load(open('$OUT/bench_c5_n2_share.json')); print(d['n_gpus'], d['config']['element_layers_per_rank'], d['rccl_ranks'], d['scaling'])"
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gputest.log
