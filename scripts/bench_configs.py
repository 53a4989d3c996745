#!/usr/bin/env python3
"""Secondary configs of BASELINE.json (C2 Hex8 Poisson 128^3, C3 Tet4 elasticity BCC res 75 permuted,
C4 Hex27 NeoHookean 50x50x80) timed on one GPU: kernel ms, elements/s, algorithmic GB/s.  Not the headline
bench (that is bench.py); results are copied under profiles/."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import fenris_amd as fa
from fenris_amd import quadrature


def timed(eng, values, flags, steps=5, warmup=2):
    for _ in range(warmup):
        eng.assemble_matrix_async(values, flags)
    eng.poll_status()
    torch.cuda.synchronize()
    ev = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.assemble_matrix_async(values, flags)
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    eng.poll_status()
    return sum(a.elapsed_time(b) for a, b in ev) / steps


def run(name, mesh, op, rule, params, u, n, d, s, scatters=("gather", "atomic", "colored")):
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    w, p = rule
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if params is not None:
        qt = qt.with_uniform_data(params)
    asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(op)
           .with_quadrature_table(qt).with_u(u).build())
    t0 = time.perf_counter()
    nnz = eng.build_pattern()
    tp = time.perf_counter() - t0
    values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    E, N = mesh.num_elements(), mesh.num_nodes()
    abytes = E * n * 4 + N * d * 8 + (s * N * 8 if u is not None else 0) + nnz * 12 + (s * N + 1) * 8
    out = {"config": name, "elements": E, "nodes": N, "nnz": nnz, "pattern_s": tp, "algorithmic_bytes": abytes, "modes": {}}
    if os.environ.get("BENCH_GATHER_ONLY"):
        scatters = ("gather",)
    for sc in scatters:
        flags = {"gather": fa.SCATTER_GATHER, "atomic": fa.SCATTER_ATOMIC, "colored": fa.SCATTER_COLORED}[sc] | fa.ASSEMBLE_OVERWRITE
        try:
            if sc == "colored":
                t0 = time.perf_counter()
                eng.color()
                out["coloring_s"] = time.perf_counter() - t0
            ms = timed(eng, values, flags)
            out["modes"][sc] = {"kernel_ms": ms, "elements_per_s": E / (ms * 1e-3), "algorithmic_GBps": abytes / (ms * 1e-3) / 1e9,
                                "frac_of_8TBps": abytes / (ms * 1e-3) / 8e12, "kernel": eng.last_kernel_name(),
                                "finite": bool(torch.isfinite(values).all())}
        except fa.FenrisError as exc:
            out["modes"][sc] = {"error": str(exc)}
    print(json.dumps(out), flush=True)
    eng.close()


def main():
    which = sys.argv[1:] or ["C2", "C3", "C4"]
    lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
    if "C2" in which:
        run("C2 Hex8 Poisson 128^3", fa.procedural.create_unit_box_uniform_hex_mesh_3d(128), fa.LaplaceOperator(),
            quadrature.tensor.hexahedron_gauss(2), None, None, 8, 3, 1)
    if "C3" in which:
        m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(75)
        rng = np.random.Generator(np.random.MT19937(12345))  # seeded permutation of vertices and elements
        vp = rng.permutation(m.num_nodes())
        inv = np.empty_like(vp)
        inv[vp] = np.arange(len(vp))
        verts = m.vertices[vp]
        conn = inv[m.connectivity.astype(np.int64)][rng.permutation(m.num_elements())].astype(np.uint64)
        run("C3 Tet4 linear elasticity BCC res 75, vertices+elements permuted (seed 12345)", fa.Mesh(verts, conn, fa.TET4),
            fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), quadrature.total_order.tetrahedron(1), lame, None, 4, 3, 3)
        if os.environ.get("BENCH_GATHER_ONLY"):
            return
        from fenris_amd import reorder
        t0 = time.perf_counter()
        mp = reorder.reorder_mesh_par(fa.Mesh(verts, conn, fa.TET4))
        t_rcm = time.perf_counter() - t0
        run(f"C3'' permuted mesh after reorder_mesh_par (reverse Cuthill-McKee on the host, {t_rcm:.1f} s)",
            mp.apply(fa.Mesh(verts, conn, fa.TET4)), fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
            quadrature.total_order.tetrahedron(1), lame, None, 4, 3, 3)
        run("C3' same mesh, generator order", m, fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
            quadrature.total_order.tetrahedron(1), lame, None, 4, 3, 3)
    if "C4" in which:
        h8 = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 5, 5, 8, 10)
        m = fa.hex27_mesh_from_hex8(h8)
        A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
        u = (0.05 * m.vertices @ A.T).reshape(-1)
        run("C4 Hex27 NeoHookean 50x50x80, hexahedron_gauss(3)", m, fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()),
            quadrature.tensor.hexahedron_gauss(3), lame, u, 27, 3, 3, scatters=("gather", "colored", "atomic"))


if __name__ == "__main__":
    main()
