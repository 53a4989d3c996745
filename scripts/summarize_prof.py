"""Summarise rocprofv3 rocpd databases (kernel stats + PMC counters) for the assembly kernels."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "stats*", "*.db"))):
    db = sqlite3.connect(f)
    print("== kernel stats:", f)
    cols = [r[1] for r in db.execute("pragma table_info(top_kernels)")]
    print("  ", cols)
    for row in db.execute("select * from top_kernels limit 12"):
        print("  ", row)
for f in sorted(glob.glob(os.path.join(root, "pmc*", "*.db"))):
    db = sqlite3.connect(f)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    print("== counters:", f)
    q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
         "where kernel_name like '%fenris_hip::k_%' group by kernel_name, counter_name") if "kernel_name" in cols else None
    if q is None:
        print("   columns:", cols)
        continue
    for k, c, v, n in db.execute(q):
        print("  %-70s %-26s per-dispatch %.6g (n=%d)" % (k.split("(")[0][-70:], c, v, n))
