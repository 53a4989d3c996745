# usage: bash scripts/gpu_profile_config.sh <config> [extra bench args]
# rocprofv3 kernel stats + PMC passes (each its own run) of `bench.py --config <config>`; summary in
# gpurun_out/prof_<config>/summary.txt (copy it to profiles/r<round>_<config>_rocprofv3_summary.txt)
CFG=${1:-ns}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$CFG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="$GRAFT_REPO_ROOT/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-secondary --no-module-warmup --no-settle $@"
# the kernel-trace pass with the set-up of the default line (device settle + placement probe, 20 timed steps): its per-kernel averages
# are the ones to compare with ms_per_step; the line it printed (hipEvent bracket, the placement it sat at) goes into the summary
BENCH_TRACE="$GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-secondary $@"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $BENCH_TRACE > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc1 -o run -- python3 $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/pmc2 -o run -- python3 $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o run -- python3 $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o run -- python3 $BENCH > $OUT/pmc4.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d $OUT/pmc5 -o run -- python3 $BENCH > $OUT/pmc5.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py gpurun_out/prof_$CFG > gpurun_out/prof_$CFG/summary.txt 2>&1
python3 - >> gpurun_out/prof_$CFG/summary.txt <<PY
import json
for ln in open("gpurun_out/prof_$CFG/stats.log"):
    if ln.startswith("{") and '"metric"' in ln:
        d = json.loads(ln)
        r = d["roofline"].get("hbm", d["roofline"])
        print("== the bench line of the kernel-trace run itself (hipEvent bracket around each step, same process as the kernel stats above):")
        print("   ms_per_step %.4f  kernel_avg_ms %.4f  kernel_min_ms %.4f  frac %.4f" % (d["ms_per_step"], r["kernel_avg_ms"], r["kernel_min_ms"], d["roofline"]["frac"]))
        print("   placement_probe:", json.dumps(d["config"].get("placement_probe")))
        print("   device_settle:", json.dumps(d["config"].get("device_settle")))
PY
# the kernel stats above average over EVERY dispatch of the process (settle, placement probe at other placements, warm-up); the timed
# steps are the last 20 dispatches of each of the configuration's kernels: their average is the one to hold against ms_per_step
python3 - >> gpurun_out/prof_$CFG/summary.txt <<PY
import glob, sqlite3
names = {"ns": ("k_affine_rows<", "k_affine_records"), "c5": ("k_affine_rows<", "k_affine_records"), "c2": ("k_affine_rows<", "k_affine_records"),
         "ns-perturbed": ("k_hex8_rows",), "c3": ("k_gather_rows_tet4",), "c4": ("k_hex27_dense_blocks", "k_rows_from_tri")}.get("$CFG", ())
for f in glob.glob("gpurun_out/prof_$CFG/stats/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    print("== the last 20 dispatches of the configuration's kernels in the kernel-trace run (= its 20 timed steps):")
    tot = 0.0
    for nm in names:
        rows = [r[0] for r in db.execute("select duration from kernels where name like ? order by start desc limit 20", ("%" + nm + "%",))]
        if rows:
            avg = sum(rows) / len(rows) / 1e6
            tot += avg
            print("   %-28s %d dispatches, average %.4f ms" % (nm, len(rows), avg))
    print("   sum %.4f ms" % tot)
PY
# keep the merge-back small: the databases stay on the box
find gpurun_out/prof_$CFG -name "*.db" -delete
cat gpurun_out/prof_$CFG/summary.txt
