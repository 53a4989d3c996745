mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
export TMPDIR=/tmp
for cfg in ns ns-perturbed c5; do
  FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/setup_$cfg.txt
done
timeout 200 python3 scripts/probe_energy_call.py 2>&1 | grep -v amdgpu.ids > $OUT/energy_call.txt
tail -40 $OUT/setup_ns.txt; cat $OUT/energy_call.txt
