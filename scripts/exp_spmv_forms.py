#!/usr/bin/env python3
"""SpMV forms inside one context (Hex8 elasticity 216^3, 19.7 GB of values): python scripts/exp_spmv_forms.py [cells]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fenris_amd as fa
from fenris_amd import quadrature
import bench
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 216
c = bench.config_problem("ns", cells, fa, quadrature, np)
mesh = c["mesh"]()
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
c["configure"](eng, mesh)
nnz = eng.build_pattern()
values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
eng.assemble_matrix(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
n = 3 * mesh.num_nodes()
x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
def t(reps=10):
    eng.spmv(values, x, y); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.spmv(values, x, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = None
for rnd in range(3):
    for label, kv in (("half_wave", {}), ("wave_per_node", {"FENRIS_HIP_SPMV_WAVE_PER_NODE": "1"})):
        for k, v in kv.items(): eng.set_option(k, v)
        ms = t()
        yy = y.clone()
        if ref is None: ref = yy
        err = float((yy - ref).abs().max() / ref.abs().max())
        print(json.dumps({"form": label, "ms": round(ms, 4), "TBps": round((nnz * 8 + nnz // 9 * 4 + 2 * n * 8) / ms / 1e9, 3), "rel_diff_vs_first": err}), flush=True)
        for k in kv: eng.set_option(k, None)
