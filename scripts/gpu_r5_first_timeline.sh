#!/bin/bash
# round 5, set-up: what runs on the device during the first assembly of a context, dispatch by dispatch (rocprofv3 kernel trace)
CFG=${1:-ns}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s/timeline_$CFG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace -o run -- python3 $GRAFT_REPO_ROOT/scripts/time_first_assembly.py $CFG > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/trace/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    # the second context: everything after the second-to-last pattern build
    t0 = rows[0][1]
    out = open("$OUT/dispatches.txt", "w")
    prev_end = t0
    for n, a, b in rows:
        out.write("%10.3f ms  +%8.3f idle  %9.3f ms  %s\n" % ((a - t0) / 1e6, (a - prev_end) / 1e6, (b - a) / 1e6, n[:110]))
        prev_end = max(prev_end, b)
    out.close()

PY
grep context $OUT/log.txt
grep -n "affine_rows\|affine_records\|hex8_rows" $OUT/dispatches.txt | head -20
