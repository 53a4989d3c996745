OUT=$GRAFT_REPO_ROOT/gpurun_out/pat; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 200 python3 $GRAFT_REPO_ROOT/scripts/time_pattern.py "$@"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/scripts/time_pattern.py "$@" > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import glob, sqlite3
for f in glob.glob("gpurun_out/pat/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for row in db.execute("select * from top_kernels limit 14"):
        print([x if not isinstance(x, str) else x[:70] for x in row])
PY
find gpurun_out/pat -name "*.db" -delete
