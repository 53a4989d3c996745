#!/bin/bash
# round 6: counters of the generic first pass (k_assemble_matrix<dump>) on Hex8 NeoHookean 128^3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_h8nh
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/scripts/bench_hex8_nh.py 128 > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $OUT/pmc1 -o run -- python3 $GRAFT_REPO_ROOT/scripts/bench_hex8_nh.py 128 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc2 -o run -- python3 $GRAFT_REPO_ROOT/scripts/bench_hex8_nh.py 128 > $OUT/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py gpurun_out/pmc_h8nh 2>&1 | grep "k_assemble_matrix<1, 3, 3>\|k_assemble_matrix<1, 4, 3>\|k_rows_from_tri" | cut -c1-60,100-200
python3 - <<PY
import glob, sqlite3
for f in glob.glob("gpurun_out/pmc_h8nh/stats/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for r in db.execute("select name, count(*), avg(duration)/1e6, min(duration)/1e6 from kernels group by name having avg(duration) > 5e5 order by avg(duration) desc limit 6"):
        print(r[0][:100], r[1], round(r[2],3), round(r[3],3))
PY
find gpurun_out/pmc_h8nh -name "*.db" -delete
