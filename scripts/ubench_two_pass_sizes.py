#!/usr/bin/env python3
"""Two-pass Hex27 NeoHookean assembly at sizes whose dense element matrices do / do not fit the 256 MB memory-side cache:
does the row gather (second pass) run faster when the element matrices were just written?  Run under
rocprofv3 --kernel-trace --stats and read the per-kernel times.    python scripts/ubench_two_pass_sizes.py n1 n2 ..."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(3)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
for n in [int(x) for x in sys.argv[1:]]:
    mesh = fa.hex27_mesh_from_hex8(fa.procedural.create_unit_box_uniform_hex_mesh_3d(n))
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    u = (0.05 * mesh.vertices @ A.T).reshape(-1)
    (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
     .with_operator(fa.MaterialEllipticOperator(fa.NeoHookeanMaterial())).with_quadrature_table(qt).with_u(u).build())
    nnz = eng.build_pattern()
    vals = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
    flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
    for _ in range(2):
        eng.assemble_matrix_async(vals, flags)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        eng.assemble_matrix_async(vals, flags)
    b.record()
    torch.cuda.synchronize()
    E = mesh.num_elements()
    print(f"n={n} E={E} dense={E * 6561 * 8 / 1e6:.0f} MB values={nnz * 8 / 1e6:.0f} MB  {a.elapsed_time(b) / 5:.3f} ms per assembly "
          f"({a.elapsed_time(b) / 5 / E * 1e3:.3f} us/element)", flush=True)
    eng.close()
    del vals
