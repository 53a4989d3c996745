"""FH_ASSEMBLE_REPRODUCIBLE: the same bits from run to run and from launch geometry to launch geometry for EVERY configuration of the gather mode,
like the reference's coloured loop (global.rs:322-373 -- each entry is a sum in a fixed order).  The row-owner kernels and the two-pass form
already are; the configurations whose one-pass kernel accumulates with LDS atomics in hardware order (Quad4 / Tri3, Hex8 with other rules or
per-point parameters: k_gather_pipelined, k_assemble_matrix<gather>) take the two-pass form under the flag."""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature
from conftest import load_golden_mesh

pytestmark = pytest.mark.gpu
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
OPS = {"LAPLACE": lambda: fa.LaplaceOperator(), "LINEAR_ELASTIC": lambda: fa.MaterialEllipticOperator(fa.LinearElasticMaterial())}
KIND = {"QUAD4": fa.QUAD4, "HEX8": fa.HEX8, "TET4": fa.TET4, "TRI3": fa.TRI3}


def _mesh(kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "TRI3":
        v, c = load_golden_mesh("square_quad4_79")
        return fa.Mesh(v, np.concatenate([c[:, [0, 1, 2]], c[:, [0, 2, 3]]]), fa.TRI3)
    m = {"QUAD4": lambda: fa.procedural.create_unit_square_uniform_quad_mesh_2d(7),
         "HEX8": lambda: fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 5),
         "TET4": lambda: fa.procedural.create_unit_box_uniform_tet_mesh_3d(3)}[kind]()
    h = 1.0 / 7
    return fa.Mesh(m.vertices + rng.uniform(-0.1 * h, 0.1 * h, m.vertices.shape), m.connectivity, m.elem_kind)
TWO_PASS = ("k_assemble_matrix<dump> + k_rows_from_dense", "k_assemble_matrix<dump> + k_rows_from_tri")   # (3 x 3 blocks on 3D elements: triangles)
FLAGS = fa.SCATTER_GATHER | fa.ASSEMBLE_REPRODUCIBLE


def _case(name):
    """(mesh, kind, op, (w, p), per-point parameters or None)"""
    if name == "quad4 elasticity":
        return _mesh("QUAD4"), "QUAD4", "LINEAR_ELASTIC", quadrature.tensor.quadrilateral_gauss(2), None
    if name == "tri3 laplace":
        return _mesh("TRI3"), "TRI3", "LAPLACE", quadrature.total_order.triangle(2), None
    if name == "hex8 general, 27 points":
        return _mesh("HEX8"), "HEX8", "LINEAR_ELASTIC", quadrature.tensor.hexahedron_gauss(3), None
    if name == "hex8 general, per-point parameters":
        w, p = quadrature.tensor.hexahedron_gauss(2)
        pairs = np.stack([LAME.mu * (1.0 + 0.03 * np.arange(len(w))), LAME.lambda_ * (1.0 - 0.02 * np.arange(len(w)))], axis=1)
        return _mesh("HEX8"), "HEX8", "LINEAR_ELASTIC", (w, p), pairs
    raise KeyError(name)


def _build(eng, oracle, mesh, kind, op, rule, pairs):
    w, p = rule
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    oparams = None
    if op != "LAPLACE":
        if pairs is None:
            qt = qt.with_uniform_data(LAME)
            oparams = LAME.as_pair()
        else:
            qt = qt.with_data([fa.LameParameters(*x) for x in pairs])
            oparams = pairs
    asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(OPS[op]()).with_quadrature_table(qt)
           .with_u(None).build())
    ref = oracle.ElementAssembler(KIND[kind], getattr(oracle, op), mesh.vertices, mesh.connectivity, w, p, params=oparams)
    return asm, ref


@pytest.mark.parametrize("name", ["quad4 elasticity", "tri3 laplace", "hex8 general, 27 points", "hex8 general, per-point parameters"])
def test_reproducible_flag_routes_the_atomic_kernels_through_two_passes(oracle, name):
    mesh, kind, op, rule, pairs = _case(name)
    first = None
    for grids in (None, ("3", "5"), ("64", "200")):
        eng = fa.Engine(0)
        try:
            if grids:
                eng.set_option("FENRIS_HIP_TWO_PASS_GRID", grids[0])
                eng.set_option("FENRIS_HIP_TWO_PASS_ROWS_GRID", grids[1])
            asm, ref = _build(eng, oracle, mesh, kind, op, rule, pairs)
            # without the flag: the one-pass kernel with LDS atomics
            k0 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
            assert eng.last_kernel_name() in ("k_gather_pipelined", "k_assemble_matrix<gather>"), eng.last_kernel_name()
            for _ in range(2):
                k = fa.CsrAssembler(FLAGS).assemble(asm)
                assert eng.last_kernel_name() in TWO_PASS
                if first is None:
                    first = k.values.copy()
                    st, _, ro, ci, vals = oracle.assemble(ref)
                    assert st == 0 and np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
                    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
                assert np.array_equal(k.values, first), (name, grids)
            assert np.abs(k0.values - first).max() <= 1e-12 * np.abs(first).max()
            # accumulating: twice the matrix, again the same bits whatever the grids
            fa.CsrAssembler(FLAGS).assemble_into_csr(k, asm)
            assert np.array_equal(k.values, first + first)
        finally:
            eng.close()


def test_reproducible_flag_leaves_the_row_owner_kernels_alone_and_rejects_atomics(oracle):
    w, p = quadrature.tensor.hexahedron_gauss(2)
    for mesh, kernel in ((fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 5), "k_affine_rows"), (_mesh("HEX8"), "k_hex8_rows"),
                         (_mesh("TET4"), "k_gather_rows")):
        eng = fa.Engine(0)
        try:
            kind = "TET4" if kernel == "k_gather_rows" else "HEX8"
            rule = quadrature.total_order.tetrahedron(2) if kind == "TET4" else (w, p)
            asm, _ = _build(eng, oracle, mesh, kind, "LINEAR_ELASTIC", rule, None)
            k = fa.CsrAssembler(FLAGS).assemble(asm)
            assert eng.last_kernel_name() == kernel
            k2 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
            assert np.array_equal(k.values, k2.values)
            with pytest.raises(fa.FenrisError):
                fa.CsrAssembler(fa.SCATTER_ATOMIC | fa.ASSEMBLE_REPRODUCIBLE).assemble(asm)
        finally:
            eng.close()
