#!/usr/bin/env python3
"""The other kernels of the path on one GPU, each with its algorithmic bytes and fraction of the 8 TB/s HBM roofline:
residual vector, energy, source vector, SpMV, one Jacobi-PCG iteration and the pattern build on the north-star mesh
(Hex8 linear elasticity 216^3), and the shapes benches/assembly.rs:126-241 defines (Poisson Tet4 on the BCC unit cube at
res 5 / 10 / 20: numeric assembly into a pre-built pattern, and the pattern itself for Poisson and 3-D elasticity).
Secondary numbers (JSON lines); the headline metric is bench.py.    python scripts/bench_other_kernels.py [cells]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

PEAK = 8000.0  # GB/s


def ev_time(fn, steps=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def line(name, ms, nbytes, **kw):
    gbps = nbytes / ms / 1e6
    out = {"kernel": name, "ms": ms, "algorithmic_bytes": nbytes, "GBps": gbps, "frac_of_8TBps": gbps / PEAK}
    out.update(kw)
    print(json.dumps(out), flush=True)


def north_star(cells):
    lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
    w, p = quadrature.tensor.hexahedron_gauss(2)
    mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(cells)
    stream = torch.cuda.current_stream().cuda_stream
    eng = fa.Engine(0, stream=stream)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
    E, N = mesh.num_elements(), mesh.num_nodes()
    n = 3 * N
    u = 1e-3 * np.sin(np.arange(n))
    asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
           .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(u).build())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nnz = eng.build_pattern()
    torch.cuda.synchronize()
    t_pat = time.perf_counter() - t0
    tag = f"Hex8 linear elasticity {cells}^3"
    # pattern: connectivity in, node-level offsets and sorted neighbour lists out (the scalar CSR is formed on the fly)
    line("pattern build (assemble_pattern, node level)", t_pat * 1e3, E * 8 * 4 + (N + 1) * 4 + nnz // 9 * 4, config=tag, one_time=True)
    values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    eng.assemble_matrix(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    f = torch.zeros(n, dtype=torch.float64, device="cuda")
    # residual f(u): connectivity + vertices + u read once, the vector written once
    ms = ev_time(lambda: eng.assemble_vector_async(f))   # enqueue only, like the stiffness benchmark; errors at the poll below
    eng.poll_status()
    line("residual vector (" + eng.last_kernel_name() + ")", ms, E * 8 * 4 + N * 3 * 8 + 2 * n * 8, config=tag, elements_per_s=E / ms * 1e3)
    ms = ev_time(lambda: eng.assemble_scalar(), steps=5)
    line("energy, assemble_scalar (" + eng.last_kernel_name() + ")", ms, E * 8 * 4 + N * 3 * 8 + n * 8, config=tag, elements_per_s=E / ms * 1e3)
    if os.environ.get("FENRIS_BENCH_OTHER_OPERATORS", "1") == "1":   # the nonlinear operators through the same passes
        for opname, mat in (("NeoHookean", fa.NeoHookeanMaterial()), ("StVK", fa.StVKMaterial())):
            e2 = fa.Engine(0, stream=stream)
            (fa.ElementEllipticAssemblerBuilder(e2).with_finite_element_space(mesh).with_operator(fa.MaterialEllipticOperator(mat))
             .with_quadrature_table(qt).with_u(u).build())
            e2.build_pattern()
            ms = ev_time(lambda: e2.assemble_vector_async(f), steps=5)
            e2.poll_status()
            line(f"residual vector, {opname} (" + e2.last_kernel_name() + ")", ms, E * 8 * 4 + N * 3 * 8 + 2 * n * 8, config=f"Hex8 {opname} {cells}^3",
                 elements_per_s=E / ms * 1e3)
            ms = ev_time(lambda: e2.assemble_scalar(), steps=5)
            line(f"energy, {opname} (" + e2.last_kernel_name() + ")", ms, E * 8 * 4 + N * 3 * 8 + n * 8, config=f"Hex8 {opname} {cells}^3",
                 elements_per_s=E / ms * 1e3)
            e2.close()
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    y = torch.zeros_like(x)
    # SpMV on the blocked CSR: values, one column index per 3 x 3 block, x gathered (counted once), y written
    ms = ev_time(lambda: eng.spmv(values, x, y))
    line("SpMV y = K x (" + eng.last_kernel_name() + "<3>)", ms, nnz * 8 + nnz // 9 * 4 + 2 * n * 8, config=tag)
    bc = np.where(mesh.vertices[:, 0] < 1e-9)[0]
    eng.apply_dirichlet_csr_dev(values, bc)
    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    b[2::3] = -1.0
    eng.apply_dirichlet_rhs_dev(b, bc)
    uu = torch.zeros(n, dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    try:
        it = eng.cg_solve(values, b, uu, 1, 1e-30, 50)  # 50 iterations, never converges to 1e-30
    except fa.CgSolveError as e:
        it = e.num_iterations
    torch.cuda.synchronize()
    # one Jacobi-PCG iteration: the SpMV plus five vector passes (r, z, p, x, diag)
    line("Jacobi-PCG iteration (SpMV + fused vector kernels)", (time.perf_counter() - t0) / max(it, 1) * 1e3,
         nnz * 8 + nnz // 9 * 4 + 2 * n * 8 + 9 * n * 8, config=tag, iterations=it)
    eng.close()
    src_eng = fa.Engine(0, stream=stream)
    (fa.ElementSourceAssemblerBuilder.new(src_eng).with_finite_element_space(mesh).with_source(fa.GravitySource([0.0, 0.0, -9.81]))
     .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.Density(1000.0))).build())
    src_eng.set_operator_for_pattern(3) if hasattr(src_eng, "set_operator_for_pattern") else None
    ms = ev_time(lambda: src_eng.assemble_source_vector(f, 3, g=[0.0, 0.0, -9.81]), steps=5)
    line("source vector, GravitySource (" + src_eng.last_kernel_name() + ")", ms, E * 8 * 4 + N * 3 * 8 + n * 8, config=tag, elements_per_s=E / ms * 1e3)
    src_eng.close()


def criterion_shapes():
    """benches/assembly.rs: serial Poisson Tet4 assemble_into_csr at res 5 / 10 / 20 (canonical 1-point rule), and
    assemble_pattern for Poisson (s = 1) and 3-D elasticity (s = 3) on the same meshes"""
    w, p = quadrature.total_order.tetrahedron(1)
    lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
    for res in (5, 10, 20):
        mesh = fa.procedural.create_unit_box_uniform_tet_mesh_3d(res)
        E, N = mesh.num_elements(), mesh.num_nodes()
        for s, op, qt in ((1, fa.LaplaceOperator(), fa.UniformQuadratureTable.from_points_and_weights(p, w)),
                          (3, fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
                           fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame))):
            eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
            ts = []
            for _ in range(5):
                # a fresh assembler on the engine drops the cached pattern
                (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(op).with_quadrature_table(qt)
                 .with_u(None).build())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nnz = eng.build_pattern()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            line(f"assemble_pattern, s = {s}", min(ts) * 1e3, E * 4 * 4 + (N + 1) * 4 + nnz // (s * s) * 4,
                 config=f"Tet4 BCC unit cube res {res} ({E} elements)", bench="benches/assembly.rs:147-241", host_inclusive=True)
            if s == 1:
                values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
                ms = ev_time(lambda: eng.assemble_matrix_async(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE), steps=20)
                line("assemble_into_csr, Poisson (" + eng.last_kernel_name() + ")", ms, E * 4 * 4 + N * 3 * 8 + nnz * 12 + (N + 1) * 8,
                     config=f"Tet4 BCC unit cube res {res} ({E} elements)", bench="benches/assembly.rs:126-145",
                     elements_per_s=E / ms * 1e3)
            eng.close()


if __name__ == "__main__":
    north_star(int(sys.argv[1]) if len(sys.argv) > 1 else 216)
    criterion_shapes()
