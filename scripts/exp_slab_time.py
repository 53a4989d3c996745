#!/usr/bin/env python3
"""time of what one rank of `bench.py --gpus 8` assembles (rank 1 of 8, masked, both launches) on one GPU, next to the unmasked N = 1 mesh"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fenris_amd as fa
from fenris_amd import quadrature, distributed as fd
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(2)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)
def configure(engine, mesh):
    return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh)
            .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build())
slab = fd.make_slab(1.0, 1, 1, 8, 216, 1, 8)
sa = fd.SlabAssembly(slab, configure, device=0, overlap=True, stream=torch.cuda.current_stream().cuda_stream, placement_tries=6)
flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
print("placement", sa.placement)
for rep in range(3):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        sa.main.assemble_matrix_rows_async(sa.values, flags, 0, sa.split)
        sa.main.assemble_matrix_async(sa.values, flags)
    b.record()
    torch.cuda.synchronize()
    print("slab rank 1 of 8: ms per step (both launches, no transfer)", a.elapsed_time(b) / 10)
