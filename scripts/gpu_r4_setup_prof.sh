OUT=$GRAFT_REPO_ROOT/gpurun_out/setup; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
FENRIS_HIP_VERBOSE=1 timeout 200 python3 $GRAFT_REPO_ROOT/scripts/time_first_assembly.py "$@" 2>&1 | grep -v amdgpu.ids | tail -30
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/scripts/time_first_assembly.py "$@" > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import glob, sqlite3
for f in glob.glob("gpurun_out/setup/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for row in db.execute("select * from top_kernels limit 22"):
        print([x if not isinstance(x, str) else x[:70] for x in row])
PY
find gpurun_out/setup -name "*.db" -delete
