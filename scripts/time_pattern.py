#!/usr/bin/env python3
"""Pattern build (assemble_pattern, node level) of Hex8 elasticity on a cells^3 box, three fresh contexts: wall time per build; under
rocprofv3 --kernel-trace --stats the per-kernel split.    python scripts/time_pattern.py [cells]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 216
lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(2)
mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(cells)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
for it in range(3):
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
    print("pattern build %d: %.2f ms (nnz %d); torch.zeros(nnz) right after it: %.2f ms" % (it, 1e3 * (t1 - t0), nnz, 1e3 * (t2 - t1)), flush=True)
    del values
    if it == 0:
        torch.cuda.empty_cache()      # the next context sees a fresh allocation of the values again
    eng.close()
