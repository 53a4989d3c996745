mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
export TMPDIR=/tmp
for cfg in ns ns-perturbed c5 c3 c2 c4; do
  FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/setup4_$cfg.txt
done
grep -h "context" $OUT/setup4_*.txt
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 > $OUT/tests_full.txt
tail -5 $OUT/tests_full.txt
