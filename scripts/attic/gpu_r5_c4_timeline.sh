#!/bin/bash
# round 5, C4: do the chunks of the overlapped two-pass assembly really run side by side?  rocprofv3 kernel trace of one short run; the
# start / end timestamps of the last step's dispatches (us relative to the first of them).
CH=${1:-8}; TH=${2:-64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_c4/timeline_${CH}_${TH}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export FENRIS_HIP_TWO_PASS_CHUNKS=$CH FENRIS_HIP_TWO_PASS_GATHER_THREADS=$TH
rocprofv3 --kernel-trace -d $OUT/trace -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --no-secondary --placement-tries 0 --no-settle > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/trace/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    rows = [(n, a, b) for n, a, b in rows if "hex27" in n or "rows_from_dense" in n]
    per_step = 2 * max(1, int("$CH")) if int("$CH") > 1 else 2
    last = rows[-per_step:]
    t0 = last[0][1]
    for n, a, b in last:
        print("%-22s start %9.1f us  end %9.1f us  (%.1f us)" % ("pass1 matrices" if "hex27" in n else "pass2 row gather", (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3))
    print("step span %.1f us" % ((max(b for _, _, b in last) - t0) / 1e3))
PY
grep ms_per_step $OUT/log.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
