mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
export TMPDIR=/tmp
for cfg in ns-perturbed c3 ns; do
  FENRIS_HIP_VERBOSE=1 timeout 300 python3 scripts/time_first_assembly.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/setup5_$cfg.txt
done
grep -h "context" $OUT/setup5_*.txt
grep "set-up" $OUT/setup5_ns-perturbed.txt | tail -10
timeout 2400 python3 -m pytest tests/test_hex8_rows.py tests/test_affine.py tests/test_gpu_parity.py tests/test_kernel_selection.py tests/test_partition.py tests/test_quadratic_elements.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30 > $OUT/tests5.txt
tail -5 $OUT/tests5.txt
timeout 600 python3 bench.py --config ns-perturbed --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-secondary --placement-tries 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ns-perturbed ms_per_step', d['ms_per_step'], d['config'].get('first_assembly_s'))"
