#!/usr/bin/env python3
"""Timings of the device-resident callers of the assembly path (one GPU): blocked-CSR SpMV, one Jacobi-PCG solve of
the clamped Hex8 elasticity system, source vector.  Secondary numbers; the headline metric is bench.py."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa
from fenris_amd import quadrature

def ev_time(fn, steps=10, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(2)
mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(cells)
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
       .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt)
       .with_u(np.zeros(3 * mesh.num_nodes())).build())
nnz = eng.build_pattern()
values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
eng.assemble_matrix(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
n = 3 * mesh.num_nodes()
x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
ms = ev_time(lambda: eng.spmv(values, x, y))
out = {"config": f"Hex8 linear elasticity {cells}^3", "rows": n, "nnz": nnz, "spmv_ms": ms,
       "spmv_GBps": (nnz * 8 + nnz / 9 * 4 + 2 * n * 8) / ms / 1e6}
bc = np.where(mesh.vertices[:, 0] < 1e-9)[0]
eng.apply_dirichlet_csr_dev(values, bc)
b = torch.zeros(n, dtype=torch.float64, device="cuda"); b[2::3] = -1.0
eng.apply_dirichlet_rhs_dev(b, bc)
u = torch.zeros(n, dtype=torch.float64, device="cuda")
t0 = time.perf_counter()
try:
    it = eng.cg_solve(values, b, u, 1, 1e-6, 2000)
    out["cg_status"] = "converged"
except fa.CgSolveError as e:
    it = e.num_iterations; out["cg_status"] = e.kind
torch.cuda.synchronize()
out["cg_iterations"] = it; out["cg_s"] = time.perf_counter() - t0; out["cg_ms_per_iteration"] = out["cg_s"] / max(it, 1) * 1e3
src_eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
src = (fa.ElementSourceAssemblerBuilder.new(src_eng).with_finite_element_space(mesh)
       .with_source(fa.GravitySource([0.0, 0.0, -9.81]))
       .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.Density(1000.0))).build())
f = torch.zeros(n, dtype=torch.float64, device="cuda")
out["source_vector_ms"] = ev_time(lambda: src_eng.assemble_source_vector(f, 3, g=[0.0, 0.0, -9.81]), steps=5)
print(json.dumps(out), flush=True)
