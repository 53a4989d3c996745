#!/bin/bash
# round 4, review item 4: per-instance L2 channel write counters of k_affine_rows for the contiguous and the chunked dealing of positions
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4/chan
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for c in 0 32; do
  FENRIS_HIP_AFFINE_CHUNK=$c rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_LEVEL -d $OUT/c$c -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config ns --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-secondary --no-module-warmup --no-settle --placement-tries 0 > $OUT/c$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import glob, sqlite3
for c in (0, 32):
    for f in glob.glob(f"gpurun_out/r4/chan/c{c}/*.db"):
        db = sqlite3.connect(f)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        print("chunk", c, "columns:", cols)
        q = "select counter_name, count(*), min(value), avg(value), max(value), sum(value) from counters_collection where kernel_name like '%k_affine_rows%' group by counter_name"
        for row in db.execute(q):
            print("   ", row)
        # per-dimension breakdown if the view has one
        # per-instance values (one row per counter instance and dispatch): spread over the L2 channels of the largest dispatches
        try:
            rows = list(db.execute("select e.event_id, e.pmc_id, e.value, e.extdata from rocpd_pmc_event e"))
            import collections, json
            by = collections.defaultdict(list)
            for ev, pmc, val, ext in rows:
                by[(ev, pmc)].append(val)
            names = {}
            for t in tabs:
                if t.startswith("rocpd_info_pmc"):
                    pc = [r[1] for r in db.execute(f"pragma table_info({t})")]
                    for r in db.execute(f"select * from {t}"):
                        d = dict(zip(pc, r))
                        names[d.get("id")] = d.get("symbol") or d.get("name")
            shown = 0
            for (ev, pmc), vals in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                if len(vals) > 1 and shown < 8:
                    import statistics
                    print(f"    event {ev} {names.get(pmc, pmc)}: {len(vals)} instances, min {min(vals):.4g} mean {statistics.mean(vals):.4g} max {max(vals):.4g} (max/mean {max(vals)/max(statistics.mean(vals),1e-9):.3f})")
                    shown += 1
            print("    rows:", len(rows), "distinct (event, counter):", len(by), "sample extdata:", rows[0][3] if rows else None)
        except Exception as exc:
            print("    per-instance query failed:", repr(exc))
PY
find gpurun_out/r4/chan -name "*.db" -delete
