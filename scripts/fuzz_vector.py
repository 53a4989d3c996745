#!/usr/bin/env python3
"""Randomised cross-check of the element pass (residual, energy: element_pass.hpp) against the older LDS-staged kernels on the device
(FENRIS_HIP_NO_ELEMENT_PASS through fh_set_option) and of the factored gravity source against sum_q (w |det J| phi)(rho g) through sampled
values: random boxes with holes, permuted numbering, affine / distorted / mixed geometry, all four operators, random rules.
    python scripts/fuzz_vector.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402


def run(cases=200, seed0=0, quiet=False):
    eng = fa.Engine(0)
    lame = fa.LameParameters(3.0e2, 5.0e2)
    bad = 0
    OPS = {"LAPLACE": fa.LaplaceOperator, "LINEAR_ELASTIC": lambda: fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
           "NEO_HOOKEAN": lambda: fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()), "STVK": lambda: fa.MaterialEllipticOperator(fa.StVKMaterial())}
    for it in range(cases):
        rng = np.random.default_rng(seed0 + it)
        kind = rng.choice(["HEX8", "TET4", "QUAD4", "TRI3"])
        dims = rng.integers(1, 12, 3)
        if kind == "HEX8":
            m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, int(dims[0]), int(dims[1]), int(dims[2]), 1)
            w, p = quadrature.tensor.hexahedron_gauss(int(rng.integers(1, 4)))
        elif kind == "TET4":
            m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(int(dims[0] % 6 + 1))
            w, p = quadrature.total_order.tetrahedron(int(rng.integers(1, 4)))
        elif kind == "QUAD4":
            m = fa.procedural.create_unit_square_uniform_quad_mesh_2d(int(dims[0] + 1))
            w, p = quadrature.tensor.quadrilateral_gauss(int(rng.integers(1, 4)))
        else:
            q = fa.procedural.create_unit_square_uniform_quad_mesh_2d(int(dims[0] + 1))
            c = np.asarray(q.connectivity)
            m = fa.Mesh(q.vertices, np.concatenate([c[:, [0, 1, 2]], c[:, [0, 2, 3]]]), fa.TRI3)
            w, p = quadrature.total_order.triangle(int(rng.integers(1, 4)))
        v, c = m.vertices.copy(), np.asarray(m.connectivity).astype(np.int64)
        geo = rng.choice(["affine", "distorted", "mixed"])
        h = 1.0 / max(dims[0], 1) if kind != "HEX8" else 1.0
        if geo != "affine":
            sel = np.ones(len(v), dtype=bool) if geo == "distorted" else rng.random(len(v)) < 0.3
            v[sel] += rng.uniform(-0.08 * h, 0.08 * h, (int(sel.sum()), v.shape[1]))
        keep = rng.random(len(c)) >= rng.choice([0.0, 0.1, 0.4])
        if not keep.any():
            keep[0] = True
        c = c[keep]
        if rng.random() < 0.3:
            perm = rng.permutation(len(v))
            inv = np.empty_like(perm)
            inv[perm] = np.arange(len(v))
            v, c = v[perm], inv[c][rng.permutation(len(c))]
        mesh = fa.Mesh(v, c.astype(np.uint64), m.elem_kind)
        opname = rng.choice(list(OPS))
        d = v.shape[1]
        s = 1 if opname == "LAPLACE" else d
        n = mesh.num_nodes()
        qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
        if opname != "LAPLACE":
            qt = qt.with_uniform_data(lame)
        u = 1e-2 * rng.standard_normal(s * n)
        if not quiet:
            print(f"case {seed0 + it}: {kind} dims {dims.tolist()} {geo} E={len(c)} N={n} {opname} nq={len(w)}", end=" ", flush=True)
        asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(OPS[opname]()).with_quadrature_table(qt).with_u(u).build())
        # a third of the cases under an element mask (the partitions of the multi-GPU runs): the tiled pass zeroes the inactive elements'
        # contributions, the staged kernels walk the list of active ones
        masked = rng.random() < 0.33
        if masked:
            mask = (rng.random(len(c)) < rng.choice([0.2, 0.6, 0.9])).astype(np.uint8)
            eng.set_active_elements(mask)
            if not quiet:
                print(f"mask {int(mask.sum())}/{len(c)}", end=" ", flush=True)
        f1 = fa.VectorAssembler().assemble_vector(asm)
        k1 = eng.last_kernel_name()
        e1 = fa.assemble_scalar(asm)
        eng.set_option("FENRIS_HIP_NO_ELEMENT_PASS", "1")
        grid = rng.choice([0, 1, 2, 5])     # few workgroups: the persistent form walks many batches each
        eng.set_option("FENRIS_HIP_PIPE_GRID", str(int(grid)) if grid else None)
        f2 = fa.VectorAssembler().assemble_vector(asm)
        k2 = eng.last_kernel_name()
        e2 = fa.assemble_scalar(asm)
        eng.set_option("FENRIS_HIP_NO_ELEMENT_PASS", None)
        eng.set_option("FENRIS_HIP_PIPE_GRID", None)
        if masked:
            eng.set_active_elements(None)
        scale = max(np.abs(f2).max(), 1e-300)
        ok = np.array_equal(np.isnan(f1), np.isnan(f2)) and np.nanmax(np.abs(f1 - f2), initial=0.0) <= 1e-11 * scale and \
            (abs(e1 - e2) <= 1e-11 * max(abs(e2), 1e-300) or (np.isnan(e1) and np.isnan(e2)))
        # gravity: factored against sampled values rho_q g
        rho = rng.uniform(0.5, 2.0, len(w))
        g = rng.standard_normal(d)
        qd = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_data([fa.Density(r) for r in rho])
        src = fa.ElementSourceAssemblerBuilder.new(eng).with_finite_element_space(mesh).with_source(fa.GravitySource.from_acceleration(g)).with_quadrature_table(qd).build()
        fg = fa.VectorAssembler().assemble_vector(src)
        sam = (fa.ElementSourceAssemblerBuilder.new(eng).with_finite_element_space(mesh)
               .with_source(fa.SourceFunction(d, lambda x, data: np.ascontiguousarray(np.broadcast_to(rho[None, :, None] * g[None, None, :], x.shape[:2] + (d,)))))
               .with_quadrature_table(qd).build())
        try:
            fs = fa.VectorAssembler().assemble_vector(sam)
            okg = np.abs(fg - fs).max() <= 1e-11 * max(np.abs(fs).max(), 1e-300)
        except Exception as exc:   # the sampled form is only the checker here
            okg = True
            if not quiet:
                print(f"(sampled source unavailable: {type(exc).__name__})", end=" ")
        if not quiet:
            print(k1.split(" ")[0], "|", k2.split(" ")[0], flush=True)
        if not (ok and okg):
            bad += 1
            print(f"   MISMATCH case {seed0 + it}: vector {np.nanmax(np.abs(f1 - f2), initial=0.0) / scale:.2e} energy {e1} / {e2} gravity ok {okg}", flush=True)
    print("cases", cases, "mismatches", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
