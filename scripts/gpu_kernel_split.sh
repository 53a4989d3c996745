# usage: bash scripts/gpu_kernel_split.sh <config> [ENV=VALUE ...]   -- rocprofv3 kernel trace of 20 timed steps of bench.py --config <config>: the
# average duration of the last 20 dispatches of every kernel that ran at least 20 times (the per-kernel split of ms_per_step)
CFG=${1:-c4}; shift
for kv in "$@"; do export "$kv"; done
OUT=$GRAFT_REPO_ROOT/gpurun_out/split_$CFG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-secondary > $OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY | tee $OUT/split.txt
import glob, sqlite3, json
for ln in open("gpurun_out/split_$CFG/stats.log"):
    if ln.startswith("{") and '"metric"' in ln:
        d = json.loads(ln); print("ms_per_step %.4f" % d["ms_per_step"])
for f in glob.glob("gpurun_out/split_$CFG/stats/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kt = "kernels" if "kernels" in tabs else None
    names = [r[0] for r in db.execute("select name from kernels group by name having count(*) >= 20")]
    for nm in names:
        rows = [r[0] for r in db.execute("select duration from kernels where name = ? order by start desc limit 20", (nm,))]
        print("   %-90s last 20: %.4f ms" % (nm[:90], sum(rows) / len(rows) / 1e6))
PY
find gpurun_out/split_$CFG -name "*.db" -delete
