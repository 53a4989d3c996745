#!/usr/bin/env python3
"""Which stiffness kernel FH_SCATTER_GATHER runs for (element, geometry, operator, quadrature rule, material data, element mask): the
library's selection written out as a table.  tests/test_kernel_selection.py holds the table this script printed and asserts it.
    python scripts/kernel_selection_table.py            # prints the rows as Python tuples"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))


def meshes():
    rng = np.random.default_rng(11)
    hexm = fa.procedural.create_unit_box_uniform_hex_mesh_3d(5)
    out = {
        "hex8 affine": hexm,
        "hex8 general": fa.Mesh(hexm.vertices + 0.02 * rng.uniform(-1, 1, hexm.vertices.shape), hexm.connectivity, fa.HEX8),
        "tet4": fa.procedural.create_unit_box_uniform_tet_mesh_3d(3),
        "quad4": fa.procedural.create_unit_square_uniform_quad_mesh_2d(6),
        "tri3": fa.procedural.create_unit_square_uniform_tri_mesh_2d(6),
        "hex27": fa.hex27_mesh_from_hex8(fa.procedural.create_unit_box_uniform_hex_mesh_3d(2)),
    }
    return out


def rules(name):
    if name.startswith("hex"):
        return {"gauss2": quadrature.tensor.hexahedron_gauss(2), "gauss3": quadrature.tensor.hexahedron_gauss(3)}
    if name == "tet4":
        return {"strength1": quadrature.total_order.tetrahedron(1), "strength2": quadrature.total_order.tetrahedron(2)}
    if name == "quad4":
        return {"gauss2": quadrature.tensor.quadrilateral_gauss(2)}
    return {"strength2": quadrature.total_order.triangle(2)}


def operators(d):
    ops = {"Laplace": (fa.LaplaceOperator(), 1)}
    ops["LinearElastic"] = (fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), d)
    ops["NeoHookean"] = (fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()), d)
    ops["StVK"] = (fa.MaterialEllipticOperator(fa.StVKMaterial()), d)
    return ops


def table():
    rows = []
    for mname, mesh in meshes().items():
        d = mesh.vertices.shape[1]
        for rname, (w, p) in rules(mname).items():
            if mname == "hex27" and rname == "gauss2":
                continue
            for oname, (op, s) in operators(d).items():
                for data in ("uniform", "per-point"):
                    if oname == "Laplace" and data != "uniform":
                        continue
                    for mask in (False, True):
                        eng = fa.Engine(0)
                        try:
                            qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
                            if oname != "Laplace":
                                if data == "uniform":
                                    qt = qt.with_uniform_data(LAME)
                                else:
                                    qt = qt.with_data([fa.LameParameters(LAME.mu * (1.0 + 0.1 * q), LAME.lambda_) for q in range(len(w))])
                            u = None if oname in ("Laplace", "LinearElastic") else 1e-3 * np.sin(np.arange(s * mesh.num_nodes()))
                            asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(op)
                                   .with_quadrature_table(qt).with_u(u).build())
                            if mask:
                                eng.set_active_elements(np.arange(mesh.num_elements()) % 4 != 1)
                            fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
                            rows.append((mname, rname, oname, data, mask, eng.last_kernel_name()))
                        finally:
                            eng.close()
    return rows


if __name__ == "__main__":
    for r in table():
        print(repr(r) + ",")
