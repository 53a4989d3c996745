#!/usr/bin/env python3
"""Residual vector of Hex8 elasticity on a cells^3 box (default 216), a few calls: the thing to put under rocprofv3 --kernel-trace --stats
for the per-kernel times of the two passes.    python scripts/time_residual.py [cells] [linear|neo|stvk] [perturbed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 216
mat = sys.argv[2] if len(sys.argv) > 2 else "linear"
lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(2)
mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(cells)
if "perturbed" in sys.argv:
    rng = np.random.Generator(np.random.MT19937(2024))
    mesh = fa.Mesh(mesh.vertices + (0.1 / cells) * rng.uniform(-1, 1, mesh.vertices.shape), mesh.connectivity, fa.HEX8)
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
material = {"linear": fa.LinearElasticMaterial, "neo": fa.NeoHookeanMaterial, "stvk": fa.StVKMaterial}[mat]()
u = 1e-3 * np.sin(np.arange(3 * mesh.num_nodes()))
(fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(fa.MaterialEllipticOperator(material))
 .with_quadrature_table(qt).with_u(u).build())
out = torch.zeros(3 * mesh.num_nodes(), dtype=torch.float64, device="cuda")
for _ in range(3):
    eng.assemble_vector(out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    eng.assemble_vector(out)
b.record()
torch.cuda.synchronize()
print("residual", mat, cells, eng.last_kernel_name(), "%.4f ms" % (a.elapsed_time(b) / 10), flush=True)
eng.close()
