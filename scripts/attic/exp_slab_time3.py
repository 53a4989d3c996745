#!/usr/bin/env python3
"""mask on / off in ONE context on the same values array (extended mesh of rank 1 of 8): is the masked sweep itself slower?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fenris_amd as fa
from fenris_amd import quadrature, distributed as fd
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(2)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)
slab = fd.make_slab(1.0, 1, 1, 8, 216, 1, 8)
flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
(fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(slab.mesh).with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
 .with_quadrature_table(qt).with_u(None).build())
nnz = eng.build_pattern()
vals = torch.zeros(nnz, dtype=torch.float64, device="cuda")
allon = np.ones(slab.mesh.num_elements(), dtype=np.uint8)
for rnd in range(2):
    for name, mask in (("no mask", None), ("all-ones mask", allon), ("slab mask", slab.active), ("no mask", None)):
        eng.set_active_elements(mask)
        t = [eng.time_assembly(vals, flags, 10) for _ in range(3)]
        eng.set_option("FENRIS_HIP_AFFINE_NO_CLEAR", "1")
        t2 = [eng.time_assembly(vals, flags, 10) for _ in range(3)]
        eng.set_option("FENRIS_HIP_AFFINE_NO_CLEAR", None)
        eng.set_option("FENRIS_HIP_AFFINE_GRID", "3072")
        t3 = [eng.time_assembly(vals, flags, 10) for _ in range(3)]
        eng.set_option("FENRIS_HIP_AFFINE_GRID", None)
        print(f"round {rnd} {name:14s} {eng.last_kernel_name():16s} ms {min(t):.3f}  without clearing {min(t2):.3f}  grid 3072 {min(t3):.3f}", flush=True)
