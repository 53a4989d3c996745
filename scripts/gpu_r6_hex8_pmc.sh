# LDS counters of k_hex8_rows with its own operand arrangement (FENRIS_HIP_ABLATE=64: instrumented instantiation, nothing off) and with the conflict-free
# stand-in (72)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_hex8; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_hex8
export TMPDIR=/tmp
cd /tmp
for ab in 64 72; do
rm -rf $OUT/c$ab; mkdir -p $OUT/c$ab
FENRIS_HIP_ABLATE=$ab rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/c$ab/pmc1 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config ns-perturbed --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-secondary --no-module-warmup --no-settle --placement-tries 0 > $OUT/c$ab/log 2>&1
done
cd $GRAFT_REPO_ROOT
for ab in 64 72; do echo "== FENRIS_HIP_ABLATE=$ab"; python3 scripts/summarize_prof.py gpurun_out/r6_hex8/c$ab 2>&1 | grep "k_hex8_rows"; done | tee -a $OUT/conflicts.txt
find $OUT -name "*.db" -delete
