timeout 1200 python3 -m pytest tests/test_affine.py tests/test_gpu_parity.py tests/test_golden_csr.py tests/test_full_size_slabs.py tests/test_source.py tests/test_bindings.py -x -q -m gpu 2>&1 | grep -v "HIP version\|ROCm version\|Hostname\|Librccl\|RCCL\|amdgpu" | tail -3
for r in 1 2 3; do for v in old new; do echo -n "$v "; FENRIS_HIP_LIB=$PWD/scripts/bin/lib_$v/libfenris_hip.so timeout 200 python3 bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['roofline'].get('kernel_avg_ms'))"; done; done
