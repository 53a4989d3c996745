# round 5: the residual's two kernels -- durations (kernel trace) and PMC of the element pass
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5res; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/kt -o run -- python3 $GRAFT_REPO_ROOT/scripts/time_residual.py "$@" > $OUT/log0.txt 2>&1
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc1 -o run -- python3 $GRAFT_REPO_ROOT/scripts/time_residual.py "$@" > $OUT/log1.txt 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o run -- python3 $GRAFT_REPO_ROOT/scripts/time_residual.py "$@" > $OUT/log2.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY > $OUT/summary.txt
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/r5res/kt/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    agg = {}
    for n, a, b in rows:
        if "element_pass" in n or "from_partials" in n or "energy" in n or "sum_partials" in n:
            agg.setdefault(n[:90], []).append((b - a) / 1e3)
    for n, v in agg.items():
        v = v[len(v) // 2:]
        print("%-92s %4d dispatches (second half), average %.1f us, min %.1f" % (n, len(v), sum(v) / len(v), min(v)))
for f in sorted(glob.glob("gpurun_out/r5res/pmc*/**/*.db", recursive=True)):
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
find gpurun_out/r5res -name "*.db" -delete
cat $OUT/summary.txt
