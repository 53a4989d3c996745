mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
export TMPDIR=/tmp
for cfg in c3 ns-perturbed; do
  FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/setup6_$cfg.txt
done
grep -h "context" $OUT/setup6_*.txt
grep "set-up" $OUT/setup6_c3.txt | tail -12
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30 > $OUT/tests6.txt
tail -4 $OUT/tests6.txt
