"""Row-owner (atomics-free) form of the stiffness kernel for Tet4 with a one-point rule (fenris_amd/csrc/rows_kernel.hpp),
the default there.  Parity against the oracle on structured, distorted, unstructured / permuted and masked meshes."""

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

pytestmark = pytest.mark.gpu
TOL = 1e-12
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))


@pytest.fixture()
def rows_engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
@pytest.mark.parametrize("name", ["bcc3", "bcc2_distorted", "sphere_permuted"])
def test_rows_kernel_tet4_matches_oracle(rows_engine, oracle, name, op):
    from conftest import load_golden_mesh

    rng = np.random.default_rng(11)
    if name == "bcc3":
        mesh = fa.procedural.create_unit_box_uniform_tet_mesh_3d(3)
    elif name == "bcc2_distorted":
        b = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
        mesh = fa.Mesh(b.vertices + 0.02 * rng.standard_normal(b.vertices.shape), b.connectivity, fa.TET4)
    else:
        v, c = load_golden_mesh("sphere_tet4_593")
        mesh = fa.Mesh(v, c[rng.permutation(len(c))], fa.TET4)
    w, p = quadrature.total_order.tetrahedron(1)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if op == "LAPLACE":
        oper, oparams, oop = fa.LaplaceOperator(), None, oracle.LAPLACE
    else:
        qt = qt.with_uniform_data(LAME)
        oper, oparams, oop = fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), LAME.as_pair(), oracle.LINEAR_ELASTIC
    asm = (fa.ElementEllipticAssemblerBuilder(rows_engine).with_finite_element_space(mesh).with_operator(oper)
           .with_quadrature_table(qt).with_u(None).build())
    ref = oracle.ElementAssembler(oracle.TET4, oop, mesh.vertices, mesh.connectivity, w, p, params=oparams)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    # the unstructured sphere has nodes with up to 40 elements: diagonal blocks of 8 lanes
    assert rows_engine.last_kernel_name() == "k_gather_rows"
    assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
    assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()
    fa.CsrAssembler(fa.SCATTER_GATHER).assemble_into_csr(k, asm)
    assert np.abs(k.values - 2.0 * vals).max() <= 2 * TOL * np.abs(vals).max()
    # element mask and a row range (what one rank of the slab partition assembles)
    active = (np.arange(mesh.num_elements()) % 4 != 2)
    rows_engine.set_active_elements(active)
    km = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert rows_engine.last_kernel_name() == "k_gather_rows"
    ka = fa.CsrAssembler(fa.SCATTER_ATOMIC).assemble(asm)
    assert np.abs(km.values - ka.values).max() <= TOL * np.abs(ka.values).max()
    rows_engine.set_active_elements(None)
    n = mesh.num_nodes()
    rows_engine.set_row_range(n // 3, 2 * n // 3)
    kr = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    s_dim = 1 if op == "LAPLACE" else 3
    lo, hi = int(ro[s_dim * (n // 3)]), int(ro[s_dim * (2 * n // 3)])
    assert np.abs(kr.values[lo:hi] - vals[lo:hi]).max() <= TOL * np.abs(vals).max()
    assert not kr.values[:lo].any() and not kr.values[hi:].any()
    rows_engine.set_row_range(0, n)


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
def test_blocks_in_locality_order_on_a_permuted_numbering(oracle, op):
    """A numbering without locality (vertices and elements randomly permuted, what bench.py --config c3 assembles): the
    owner blocks are formed in Morton order of the coordinates and the lanes store to the real rows.  Same pattern, same
    values as the oracle; with an element mask; and after switching to an operator the row-owner kernel does not serve
    the context falls back to blocks in node order."""
    b = fa.procedural.create_unit_box_uniform_tet_mesh_3d(6)
    rng = np.random.Generator(np.random.MT19937(7))
    vp = rng.permutation(b.num_nodes())
    inv = np.empty_like(vp)
    inv[vp] = np.arange(len(vp))
    conn = inv[b.connectivity.astype(np.int64)][rng.permutation(b.num_elements())].astype(np.uint64)
    mesh = fa.Mesh(b.vertices[vp], conn, fa.TET4)
    w, p = quadrature.total_order.tetrahedron(1)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if op == "LAPLACE":
        oper, oparams, oop = fa.LaplaceOperator(), None, oracle.LAPLACE
    else:
        qt = qt.with_uniform_data(LAME)
        oper, oparams, oop = fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), LAME.as_pair(), oracle.LINEAR_ELASTIC
    ref = oracle.ElementAssembler(oracle.TET4, oop, mesh.vertices, mesh.connectivity, w, p, params=oparams)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    import os
    results = {}
    for forced in ("0", "1"):    # by the locality measure itself, and forced (the measure already says yes here)
        os.environ["FENRIS_HIP_NODE_ORDER"] = forced
        os.environ["FENRIS_HIP_VERBOSE"] = "1"
        try:
            eng = fa.Engine(0)
        finally:
            del os.environ["FENRIS_HIP_NODE_ORDER"], os.environ["FENRIS_HIP_VERBOSE"]
        asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(oper)
               .with_quadrature_table(qt).with_u(None).build())
        k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert eng.last_kernel_name() == "k_gather_rows"
        assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
        assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()
        fa.CsrAssembler(fa.SCATTER_GATHER).assemble_into_csr(k, asm)          # accumulate
        assert np.abs(k.values - 2.0 * vals).max() <= 2 * TOL * np.abs(vals).max()
        active = (np.arange(mesh.num_elements()) % 5 != 1)
        eng.set_active_elements(active)
        km = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        ka = fa.CsrAssembler(fa.SCATTER_ATOMIC).assemble(asm)
        assert np.abs(km.values - ka.values).max() <= TOL * np.abs(ka.values).max()
        eng.set_active_elements(None)
        n = mesh.num_nodes()
        eng.set_row_range(n // 4, n // 2)                                       # a row range: blocks in node order again
        kr = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        s_dim = 1 if op == "LAPLACE" else 3
        lo, hi = int(ro[s_dim * (n // 4)]), int(ro[s_dim * (n // 2)])
        assert np.abs(kr.values[lo:hi] - vals[lo:hi]).max() <= TOL * np.abs(vals).max()
        assert not kr.values[:lo].any() and not kr.values[hi:].any()
        eng.set_row_range(0, n)
        results[forced] = k.values
        eng.close()
    # without the locality order (FENRIS_HIP_NO_NODE_ORDER): the same values up to the order of the sums
    os.environ["FENRIS_HIP_NO_NODE_ORDER"] = "1"
    try:
        eng = fa.Engine(0)
    finally:
        del os.environ["FENRIS_HIP_NO_NODE_ORDER"]
    asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(oper)
           .with_quadrature_table(qt).with_u(None).build())
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()
    eng.close()
