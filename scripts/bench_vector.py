#!/usr/bin/env python3
"""Residual-vector kernel timing (Hex8 linear elasticity, one GPU):  python scripts/bench_vector.py [cells]"""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa
from fenris_amd import quadrature

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(2)
mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(cells)
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
u = torch.from_numpy((0.05 * mesh.vertices @ A.T).reshape(-1)).cuda()
asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
       .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(u).build())
f = torch.zeros(3 * mesh.num_nodes(), dtype=torch.float64, device="cuda")
for _ in range(2): eng.assemble_vector(f)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.assemble_vector(f); b.record(); ts.append((a, b))
torch.cuda.synchronize()
ms = sum(a.elapsed_time(b) for a, b in ts) / len(ts)
print(json.dumps({"cells": cells, "elements": mesh.num_elements(), "vector_ms": ms, "elements_per_s": mesh.num_elements() / ms * 1e3,
                  "kernel": eng.last_kernel_name(), "ablate": os.environ.get("FENRIS_HIP_ABLATE", "")}))
