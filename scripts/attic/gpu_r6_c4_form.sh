# round 6: the one-operand block form of C4's first pass (hex27_blocks.hpp) against the 16 x 16 tiles -- parity tests, in-context A/B, phase trace
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_c4; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_c4
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_hex27_mfma.py tests/test_quadratic_elements.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15
timeout 600 python3 scripts/ab_in_context.py --config c4 --rounds 5 "tiles:FENRIS_HIP_HEX27_FORM=0" "blocks2:FENRIS_HIP_HEX27_FORM=2" "blocks2_3wg:FENRIS_HIP_HEX27_FORM=2,FENRIS_HIP_HEX27_WGS_PER_CU=3" 2>&1 | grep -v "amdgpu.ids" | tee $OUT/ab.txt
for f in 2 0; do
FENRIS_HIP_HEX27_FORM=$f FENRIS_HIP_TRACE=1 timeout 300 python3 bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-secondary --placement-tries 0 --no-settle 2>&1 | grep -i "trace\|ms_per_step" | cut -c1-200 | tail -12 | tee $OUT/trace_form$f.txt
done
