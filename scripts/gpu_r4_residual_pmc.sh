# PMC of the residual's element pass (scripts/time_residual.py): instruction counts and VALU utilisation
OUT=$GRAFT_REPO_ROOT/gpurun_out/vtpmc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc1 -o run -- python3 $GRAFT_REPO_ROOT/scripts/time_residual.py "$@" > $OUT/log1.txt 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o run -- python3 $GRAFT_REPO_ROOT/scripts/time_residual.py "$@" > $OUT/log2.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/vtpmc/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    t = [x for x in tabs if x.startswith("counters_collection")]
    if not t:
        print(f, "no counters table", tabs[:8]); continue
    cols = [r[1] for r in db.execute(f"pragma table_info({t[0]})")]
    kn = "kernel_name" if "kernel_name" in cols else cols[0]
    q = f"select {kn}, counter_name, sum(value), count(*) from {t[0]} group by {kn}, counter_name"
    for k, c, v, n in db.execute(q):
        if "element_pass" in k or "from_partials" in k:
            print(k[:70].split("(")[0][-40:], c, "%.4g" % (v / max(n, 1)), "per launch x", n)
PY
find gpurun_out/vtpmc -name "*.db" -delete
