#!/usr/bin/env python3
"""Timings of the other kernels on the path (one GPU): residual vector and energy for Hex8 216^3, NeoHookean
Hex8 matrix, per-point-parameter LinearElastic matrix."""
import json, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa
from fenris_amd import quadrature

def ev_time(fn, steps=5, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ts.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ts) / steps

lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(2)
for cells, opname in ((216, "LINEAR_ELASTIC"), (128, "NEO_HOOKEAN"), (128, "LINEAR_ELASTIC_PER_POINT"), (128, "STVK")):
    mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(cells)
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if opname == "LINEAR_ELASTIC_PER_POINT":
        qt = qt.with_data([fa.LameParameters(lame.mu * (1 + 0.01 * q), lame.lambda_) for q in range(8)])
        op = fa.MaterialEllipticOperator(fa.LinearElasticMaterial())
    else:
        qt = qt.with_uniform_data(lame)
        op = fa.MaterialEllipticOperator({"LINEAR_ELASTIC": fa.LinearElasticMaterial, "NEO_HOOKEAN": fa.NeoHookeanMaterial, "STVK": fa.StVKMaterial}[opname]())
    A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
    u = torch.from_numpy((0.05 * mesh.vertices @ A.T).reshape(-1)).cuda()
    asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(op).with_quadrature_table(qt).with_u(u).build())
    E = mesh.num_elements()
    out = {"config": f"Hex8 {opname} {cells}^3", "elements": E}
    nnz = eng.build_pattern()
    values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    for name, flags in (("gather", fa.SCATTER_GATHER), ("atomic", fa.SCATTER_ATOMIC)):
        if cells == 216 and name == "atomic": continue
        ms = ev_time(lambda: eng.assemble_matrix_async(values, flags | fa.ASSEMBLE_OVERWRITE))
        eng.poll_status()
        out[f"matrix_{name}_ms"] = ms; out[f"matrix_{name}_elem_per_s"] = E / ms * 1e3; out[f"matrix_{name}_kernel"] = eng.last_kernel_name()
    f = torch.zeros(3 * mesh.num_nodes(), dtype=torch.float64, device="cuda")
    t0 = time.perf_counter(); eng.assemble_vector(f); torch.cuda.synchronize(); t1 = time.perf_counter()
    ms = ev_time(lambda: eng.assemble_vector(f))
    out["vector_ms"] = ms; out["vector_elem_per_s"] = E / ms * 1e3
    t0 = time.perf_counter(); e = eng.assemble_scalar(); out["scalar_s"] = time.perf_counter() - t0; out["energy"] = e
    print(json.dumps(out), flush=True)
    eng.close(); del values, f
