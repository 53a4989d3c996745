#!/usr/bin/env python3
"""Experiment: one fresh process, one context, the headline assembly timed; does the ORDER of the big allocations decide the mode?
    python scripts/exp_fresh_order.py [values_first]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

values_first = len(sys.argv) > 1 and sys.argv[1] == "values_first"
cells = 216
lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
stream = torch.cuda.current_stream().cuda_stream
values = None
if values_first:
    values = torch.zeros(9 * (3 * cells + 1) ** 3, dtype=torch.float64, device="cuda")
mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, cells)
w, p = quadrature.tensor.hexahedron_gauss(2)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
eng = fa.Engine(0, stream=stream)
fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build()
nnz = eng.build_pattern()
if values is None:
    values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
assert values.numel() == nnz
flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
for _ in range(3):
    eng.assemble_matrix_async(values, flags)
torch.cuda.synchronize()
ts = []
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        eng.assemble_matrix_async(values, flags)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 5)
print(json.dumps({"values_first": values_first, "recs_late": bool(os.environ.get("FENRIS_HIP_RECS_LATE")), "ms": round(sorted(ts)[len(ts) // 2], 4)}), flush=True)
