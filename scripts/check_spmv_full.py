#!/usr/bin/env python3
"""SpMV on the blocked CSR at a size where every wavefront walks many nodes (Hex8 elasticity 128^3 and a distorted 96^3; Tet4 BCC 40):
fh_spmv against rocSPARSE through torch.sparse_csr on the same values, and one Jacobi-PCG solve whose residual is recomputed that way.
    python scripts/check_spmv_full.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e3, 0.3))
ok = True
for name, mesh, rule in (("Hex8 128^3", fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 128), quadrature.tensor.hexahedron_gauss(2)),
                         ("Tet4 BCC 40", fa.procedural.create_unit_box_uniform_tet_mesh_3d(40), quadrature.total_order.tetrahedron(1))):
    eng = fa.Engine(0)
    w, p = rule
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)
    asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
           .with_quadrature_table(qt).with_u(None).build())
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm, device_values=True)
    n = 3 * mesh.num_nodes()
    bc = np.where(mesh.vertices[:, 0] < 1e-9)[0]
    fa.apply_homogeneous_dirichlet_bc_csr(k, bc, 3, asm)
    ro = torch.as_tensor(np.asarray(k.row_offsets).astype(np.int64), device="cuda")
    ci = torch.as_tensor(np.asarray(k.col_indices).astype(np.int64), device="cuda")
    A = torch.sparse_csr_tensor(ro, ci, k.values, size=(n, n))
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    y = torch.zeros_like(x)
    eng.spmv(k.values, x, y)
    kern = eng.last_kernel_name()
    yr = A @ x
    e1 = ((y - yr).abs().max() / yr.abs().max()).item()
    b = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    b.view(-1, 3)[torch.as_tensor(bc, device="cuda")] = 0.0
    sol = torch.zeros_like(b)
    cg = (fa.ConjugateGradient.new().with_operator(k, asm).with_preconditioner(fa.JacobiPreconditioner()).with_max_iter(3000)
          .with_stopping_criterion(fa.RelativeResidualCriterion(1e-8)))
    it = cg.solve_with_guess(b, sol)
    res = ((b - A @ sol).norm() / b.norm()).item()
    good = e1 <= 1e-13 and res <= 2e-8
    ok &= good
    print(f"{name}: {mesh.num_elements()} elements, nnz {k.values.numel()}, {kern}: |y - y_rocsparse| / |y| = {e1:.2e}; PCG {it} iterations, "
          f"|b - A x| / |b| recomputed = {res:.2e}  {'OK' if good else 'FAILED'}", flush=True)
    eng.close()
    del A, ro, ci, k
    torch.cuda.empty_cache()
sys.exit(0 if ok else 1)
