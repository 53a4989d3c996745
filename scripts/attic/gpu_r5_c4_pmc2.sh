# PMC of C4's first pass: tiles (FENRIS_HIP_HEX27_BLOCKS unset) against blocks (=1)
export TMPDIR=/tmp
for v in 0 1; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_c4/pmc_form$v; rm -rf $OUT; mkdir -p $OUT
cd /tmp
FENRIS_HIP_HEX27_BLOCKS=$v rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-secondary --no-module-warmup --no-settle --placement-tries 0 > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
echo "=== FENRIS_HIP_HEX27_BLOCKS=$v"
python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/pmc/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = f"select s.kernel_name, i.name, avg(e.value), count(*) from {pc} e join {pi} i on e.pmc_id = i.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, i.name"
    for n, c, v, k in db.execute(q):
        if "hex27" in n: print(n[:48], c, "%.4g" % v, k)
PY
find $OUT -name "*.db" -delete
done
