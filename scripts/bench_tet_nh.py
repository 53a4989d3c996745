#!/usr/bin/env python3
"""Tet4 NeoHookean (BCC res 75, one-point rule, 5.06 M elements): stiffness (two-pass), residual, energy."""
import json, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa
from fenris_amd import quadrature

mesh = fa.procedural.create_unit_box_uniform_tet_mesh_3d(75)
lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.total_order.tetrahedron(1)
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
u = torch.from_numpy((0.05 * mesh.vertices @ A.T).reshape(-1)).cuda()
asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
       .with_operator(fa.MaterialEllipticOperator(fa.NeoHookeanMaterial())).with_quadrature_table(qt).with_u(u).build())
nnz = eng.build_pattern()
values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
f = torch.zeros(3 * mesh.num_nodes(), dtype=torch.float64, device="cuda")

def ev(fn, steps=5, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ts.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ts) / steps

out = {"config": "Tet4 NeoHookean BCC res 75 (5062500 elements)"}
out["matrix_gather_ms"] = ev(lambda: eng.assemble_matrix_async(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE))
eng.poll_status()
out["matrix_kernel"] = eng.last_kernel_name()
out["vector_ms"] = ev(lambda: eng.assemble_vector(f))
t0 = time.perf_counter(); e = eng.assemble_scalar(); out["scalar_s"] = time.perf_counter() - t0; out["energy"] = e
print(json.dumps(out))
