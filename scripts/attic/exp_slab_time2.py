#!/usr/bin/env python3
"""where a masked slab loses against the unmasked mesh: the same extended mesh (216 x 216 x 218 cells) without a mask, with the mask, with the
mask and the row range of SlabAssembly -- ms per assembly inside each context (fh_time_assembly_dev)"""
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
for name in (sys.argv[1:] or ["mask", "no mask", "mask + row range [split, n)", "no mask"]):
    if "DEDUPE" in name:
        os.environ["FENRIS_HIP_NO_LANE_DEDUPE"] = "1"
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(slab.mesh).with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
     .with_quadrature_table(qt).with_u(None).build())
    if name != "no mask":
        eng.set_active_elements(slab.active)
    nnz = eng.build_pattern()
    if "row range" in name:
        eng.set_row_range(int(slab.send_nodes[1]), slab.mesh.num_nodes())
    vals = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    eng.set_option("FENRIS_HIP_VERBOSE", "1")
    t = [eng.time_assembly(vals, flags, 10) for _ in range(4)]
    eng.set_option("FENRIS_HIP_VERBOSE", None)
    eng.set_option("FENRIS_HIP_AFFINE_NO_CARRY", "1")
    t2 = [eng.time_assembly(vals, flags, 10) for _ in range(3)]
    eng.set_option("FENRIS_HIP_AFFINE_NO_CARRY", None)
    print(f"{name:34s} {eng.last_kernel_name():20s} ms {min(t):.3f} (of {[round(x, 3) for x in t]}); without carries {min(t2):.3f}", flush=True)
    eng.close()
    del vals
    torch.cuda.empty_cache()
