#!/usr/bin/env python3
"""Hex8 linear elasticity with a piecewise-constant material (CompactQuadratureTable, one rule per material):
python scripts/bench_multimaterial.py [cells]"""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa
from fenris_amd import quadrature

tet = len(sys.argv) > 1 and sys.argv[1] == "tet"   # "tet": the C3 mesh (BCC res 75) with four materials
cells = 75 if tet else (int(sys.argv[1]) if len(sys.argv) > 1 else 128)
if tet:
    w, p = quadrature.total_order.tetrahedron(1)
    mesh = fa.procedural.create_unit_box_uniform_tet_mesh_3d(cells)
else:
    w, p = quadrature.tensor.hexahedron_gauss(2)
    mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(cells)
E = mesh.num_elements()
rules = [[fa.LameParameters(4e5 * (1 + r), 3e5 / (1 + r))] * len(w) for r in range(4)]
emap = (np.arange(E) * 2654435761 % 4).astype(np.uint64)
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
qt = fa.CompactQuadratureTable.from_quadrature_rules_and_map(p, w, rules, emap)
asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
       .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build())
nnz = eng.build_pattern()
values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
run = lambda: eng.assemble_matrix_async(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
for _ in range(2): run()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); ts.append((a, b))
torch.cuda.synchronize()
eng.poll_status()
ms = sum(a.elapsed_time(b) for a, b in ts) / len(ts)
print(json.dumps({"config": f"{'Tet4 BCC res' if tet else 'Hex8'} linear elasticity {cells}{'' if tet else '^3'}, 4 materials (compact table, rules constant over points)",
                  "elements": E, "matrix_gather_ms": ms, "elements_per_s": E / ms * 1e3, "kernel": eng.last_kernel_name()}))
