# round 6: general Hex8 (k_hex8_rows, ns-perturbed 216^3) -- what a conflict-free arrangement of the operand reads could buy at most: the instrumented
# instantiation with FENRIS_HIP_ABLATE bit 8 (sixteen consecutive vectors per sixteen lanes: NO bank conflicts, wrong sums -- timing only), and the
# other phases off one by one (2 no phase C, 4 no phase B), inside one context; then the LDS counters of the two arrangements
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_hex8; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_hex8
export TMPDIR=/tmp
V="prod: dbg:FENRIS_HIP_ABLATE=64 conflict_free:FENRIS_HIP_ABLATE=72 noC:FENRIS_HIP_ABLATE=66 noB:FENRIS_HIP_ABLATE=68 noB_noC:FENRIS_HIP_ABLATE=70 noB_conflict_free:FENRIS_HIP_ABLATE=76"
timeout 900 python3 scripts/ab_in_context.py --config ns-perturbed --rounds 3 --reps 3 $V 2>&1 | grep -v "amdgpu.ids\|trace\]" | tee $OUT/conflicts.txt
cd /tmp
for ab in 64 72; do
FENRIS_HIP_ABLATE=$ab rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_$ab -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config ns-perturbed --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-secondary --no-module-warmup --no-settle --placement-tries 0 > $OUT/pmc_$ab.log 2>&1
done
cd $GRAFT_REPO_ROOT
for ab in 64 72; do echo "== FENRIS_HIP_ABLATE=$ab"; python3 scripts/summarize_prof.py gpurun_out/r6_hex8/pmc_$ab 2>&1 | grep "k_hex8_rows"; mkdir -p gpurun_out/r6_hex8/pmc_$ab/pmc1; done | tee -a $OUT/conflicts.txt
find $OUT -name "*.db" -delete
