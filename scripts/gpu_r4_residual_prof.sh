# per-kernel times of the residual's passes: rocprofv3 kernel trace of scripts/time_residual.py (bounded: the trace of a long script takes minutes)
OUT=$GRAFT_REPO_ROOT/gpurun_out/vt; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/scripts/time_residual.py "$@" > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
cat $OUT/log.txt | grep residual
python3 - <<PY
import glob, sqlite3
for f in glob.glob("gpurun_out/vt/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for row in db.execute("select * from top_kernels limit 8"):
        print([x if not isinstance(x, str) else x[:60] for x in row])
PY
find gpurun_out/vt -name "*.db" -delete
