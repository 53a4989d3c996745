#!/usr/bin/env python3
"""First assembly of a context (owner tables, lane tables, ... built once per pattern) against the steady state, Hex8 elasticity cells^3.
Under rocprofv3 --kernel-trace --stats: the device part of the set-up by kernel.    python scripts/time_first_assembly.py [cells] [perturbed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 216
lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(2)
mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(cells)
if "perturbed" in sys.argv:
    rng = np.random.Generator(np.random.MT19937(2024))
    mesh = fa.Mesh(mesh.vertices + (0.1 / cells) * rng.uniform(-1, 1, mesh.vertices.shape), mesh.connectivity, fa.HEX8)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
for it in range(2):
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
     .with_quadrature_table(qt).with_u(None).build())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nnz = eng.build_pattern()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    eng.assemble_matrix(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    eng.assemble_matrix(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("context %d: pattern %.1f ms, values alloc %.1f ms, first assembly %.1f ms (%s), second %.2f ms" %
          (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), eng.last_kernel_name(), 1e3 * (t4 - t3)), flush=True)
    del values
    eng.close()
