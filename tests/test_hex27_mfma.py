"""Two-pass owner-computes assembly of Hex27 (dense element matrices on the fp64 matrix cores, then row gather):
parity with the oracle, NaN / singular semantics, accumulation, element mask.  GPU only."""
import numpy as np
import pytest

import fenris_amd as fa

HEX27_TWO_PASS = ("k_hex27_dense_blocks + k_rows_from_tri",)
from fenris_amd import quadrature

pytestmark = pytest.mark.gpu
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
OPS = {"LINEAR_ELASTIC": fa.LinearElasticMaterial, "NEO_HOOKEAN": fa.NeoHookeanMaterial, "STVK": fa.StVKMaterial}


@pytest.fixture(scope="module")
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


def _mesh(seed=0, cells=(2, 2, 3)):
    rng = np.random.default_rng(seed)
    m8 = fa.procedural.create_rectangular_uniform_hex_mesh(0.5, *cells, 1)
    m8 = fa.Mesh(m8.vertices + rng.uniform(-0.04, 0.04, m8.vertices.shape), m8.connectivity, m8.elem_kind)
    return fa.hex27_mesh_from_hex8(m8)


def _build(engine, oracle, mesh, op, u, per_point=False):
    w, p = quadrature.tensor.hexahedron_gauss(3)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if per_point:
        params = np.array([[LAME.mu * (1 + 0.02 * q), LAME.lambda_ * (1 - 0.01 * q)] for q in range(len(w))])
        qt = qt.with_data([fa.LameParameters(*x) for x in params])
    else:
        params = np.array(LAME.as_pair())
        qt = qt.with_uniform_data(LAME)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh)
           .with_operator(fa.MaterialEllipticOperator(OPS[op]())).with_quadrature_table(qt).with_u(u).build())
    ref = oracle.ElementAssembler(oracle.HEX27, getattr(oracle, op), mesh.vertices, mesh.connectivity, w, p, params=params, u=u)
    return asm, ref


@pytest.mark.parametrize("op,per_point", [("LINEAR_ELASTIC", False), ("NEO_HOOKEAN", False), ("NEO_HOOKEAN", True),
                                          ("LINEAR_ELASTIC", True)])
def test_mfma_path_matches_oracle(engine, oracle, op, per_point):
    mesh = _mesh(1)
    u = 0.01 * np.random.default_rng(2).standard_normal(3 * mesh.num_nodes())
    asm, ref = _build(engine, oracle, mesh, op, u, per_point)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert engine.last_kernel_name() in HEX27_TWO_PASS
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0 and np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
    # exactly symmetric, like clone_upper_to_lower leaves every element matrix (util.rs:38-51): the matrix-core pass forms
    # the components on and above the diagonal only and writes the others as transposed copies; the row gather adds the
    # element contributions of (I, J) and (J, I) in the same (ascending element) order
    a = k.to_scipy()
    d = (a - a.T).tocoo()
    assert d.nnz == 0 or not np.any(d.data != 0.0)


def test_generic_first_pass_for_other_operators(engine, oracle):
    mesh = _mesh(3, cells=(1, 2, 2))
    u = 0.01 * np.random.default_rng(4).standard_normal(3 * mesh.num_nodes())
    asm, ref = _build(engine, oracle, mesh, "STVK", u)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert engine.last_kernel_name() == "k_assemble_matrix<dump> + k_rows_from_tri"
    vals = oracle.assemble(ref)[4]
    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()


def test_mfma_path_accumulates_and_overwrites(engine, oracle):
    import torch

    mesh = _mesh(5, cells=(1, 1, 2))
    u = 0.005 * np.random.default_rng(6).standard_normal(3 * mesh.num_nodes())
    asm, ref = _build(engine, oracle, mesh, "NEO_HOOKEAN", u)
    vals = oracle.assemble(ref)[4]
    nnz = engine.build_pattern()
    v = torch.full((nnz,), 2.0, dtype=torch.float64, device="cuda")
    engine.assemble_matrix(v, fa.SCATTER_GATHER)  # assemble_into_csr accumulates (global.rs:133-182)
    assert np.abs(v.cpu().numpy() - (vals + 2.0)).max() <= 1e-12 * np.abs(vals).max()
    engine.assemble_matrix(v, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    assert np.abs(v.cpu().numpy() - vals).max() <= 1e-12 * np.abs(vals).max()


def test_mfma_path_nan_positions_for_inverted_element(engine, oracle):
    # fenris-solid/src/materials.rs:298-300: J <= 0 => all-NaN blocks, not an error
    mesh = _mesh(7)
    u = 0.002 * np.random.default_rng(8).standard_normal(3 * mesh.num_nodes())
    nodes = mesh.connectivity[4].astype(int)
    centre = mesh.vertices[nodes].mean(axis=0)
    for n in nodes:  # reflect element 4 through its centre: F = -1.2 I there
        u[3 * n: 3 * n + 3] = -2.2 * (mesh.vertices[n] - centre)
    asm, ref = _build(engine, oracle, mesh, "NEO_HOOKEAN", u)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert engine.last_kernel_name() in HEX27_TWO_PASS
    vals = oracle.assemble(ref)[4]
    assert np.isnan(vals).any() and np.array_equal(np.isnan(k.values), np.isnan(vals))
    ok = ~np.isnan(vals)
    assert np.abs(k.values[ok] - vals[ok]).max() <= 1e-12 * np.abs(vals[ok]).max()


def test_mfma_path_singular_jacobian(engine, oracle):
    mesh = _mesh(9, cells=(1, 2, 2))
    v = mesh.vertices.copy()
    v[mesh.connectivity[2].astype(int)[:8]] = v[int(mesh.connectivity[2][0])]  # collapse the corners: det J == 0
    bad = fa.Mesh(v, mesh.connectivity, mesh.elem_kind)
    asm, _ = _build(engine, oracle, bad, "LINEAR_ELASTIC", np.zeros(3 * bad.num_nodes()))
    with pytest.raises(fa.SingularJacobianError) as ei:
        fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert ei.value.element == 2


def test_mfma_path_with_element_mask(engine, oracle):
    mesh = _mesh(11)
    u = 0.01 * np.random.default_rng(12).standard_normal(3 * mesh.num_nodes())
    asm, ref = _build(engine, oracle, mesh, "NEO_HOOKEAN", u)
    mask = (np.arange(mesh.num_elements()) % 3 != 1).astype(np.uint8)
    engine.set_active_elements(mask)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    engine.set_active_elements(None)
    ro, ci = oracle.pattern_for(ref)
    w, p = quadrature.tensor.hexahedron_gauss(3)
    sub = oracle.ElementAssembler(oracle.HEX27, oracle.NEO_HOOKEAN, mesh.vertices, mesh.connectivity[mask.astype(bool)], w, p,
                                  params=np.array(LAME.as_pair()), u=u)
    vals = np.zeros(len(ci))
    st, _ = oracle.assemble_into_csr(sub, ro, ci, vals)
    assert st == 0 and np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()


def test_two_pass_path_is_bit_reproducible(oracle):
    """the reference's coloured assembly gives the same bits every run (global.rs:322-373); so does the two-pass path: every addition into a
    node's rows comes from ONE wavefront in program order (LDS operations of a wavefront execute in order), whatever the launch grids and
    however often it runs -- Hex27 NeoHookean (C4's kernels) on a distorted mesh, four runs under three grids of the row pass"""
    mesh = _mesh(7, cells=(3, 3, 4))
    u = 0.01 * np.random.default_rng(3).standard_normal(3 * mesh.num_nodes())
    ref_vals = None
    for grid in (None, "7", "200"):
        eng = fa.Engine(0)
        try:
            if grid:
                eng.set_option("FENRIS_HIP_TWO_PASS_ROWS_GRID", grid)
            asm, _ = _build(eng, oracle, mesh, "NEO_HOOKEAN", u)
            for _ in range(2 if grid else 4):
                k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
                assert eng.last_kernel_name() in HEX27_TWO_PASS
                if ref_vals is None:
                    ref_vals = k.values.copy()
                assert np.array_equal(k.values, ref_vals), grid
        finally:
            eng.close()


@pytest.mark.parametrize("op,per_point", [("LINEAR_ELASTIC", False), ("NEO_HOOKEAN", False), ("NEO_HOOKEAN", True)])
def test_roles_rotate_and_triangles_are_symmetric(oracle, op, per_point):
    """hex27_blocks.hpp: the element matrices from 4 x 4 x 4 blocks (v_mfma_f64_4x4x4_4b), stored as upper node-block triangles -- with MANY
    elements per workgroup (12 elements on three workgroups: every wavefront takes every role, scratch of the per-point chain is reused from
    element to element), the oracle's matrix at 1e-12, symmetric bit for bit (both halves of a symmetric pair are sums of the same stored
    doubles), the same bits from run to run, and against the generic first pass with full column-major matrices in the same context."""
    eng = fa.Engine(0)
    try:
        eng.set_option("FENRIS_HIP_TWO_PASS_GRID", 3)
        mesh = _mesh(1)
        u = 0.01 * np.random.default_rng(2).standard_normal(3 * mesh.num_nodes())
        asm, ref = _build(eng, oracle, mesh, op, u, per_point)
        k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert eng.last_kernel_name() in HEX27_TWO_PASS
        st, _, ro, ci, vals = oracle.assemble(ref)
        assert st == 0 and np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
        assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
        a = k.to_scipy()
        d = (a - a.T).tocoo()
        assert d.nnz == 0 or not np.any(d.data != 0.0)
        k2 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert np.array_equal(k.values, k2.values)
        eng.set_option("FENRIS_HIP_NO_MFMA", 1)
        k3 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert eng.last_kernel_name() == "k_assemble_matrix<dump> + k_rows_from_tri"
        assert np.abs(k3.values - k.values).max() <= 1e-12 * np.abs(vals).max()
        eng.set_option("FENRIS_HIP_NO_MFMA", None)
        # the two passes work on the element's nodes in lexicographic order of their reference positions (engine_two_pass.hip); on the element's
        # own order (the A/B switch) the sum of grad u runs over the nodes in another order: the same matrix to rounding, symmetric bit for bit,
        # and the tables of either order are rebuilt when the switch changes inside one context
        eng.set_option("FENRIS_HIP_HEX27_NO_LEX", 1)
        k4 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert eng.last_kernel_name() in HEX27_TWO_PASS
        assert np.abs(k4.values - vals).max() <= 1e-12 * np.abs(vals).max()
        assert np.abs(k4.values - k.values).max() <= 1e-13 * np.abs(vals).max()
        d4 = (k4.to_scipy() - k4.to_scipy().T).tocoo()
        assert d4.nnz == 0 or not np.any(d4.data != 0.0)
        eng.set_option("FENRIS_HIP_HEX27_NO_LEX", None)
        k5 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert np.array_equal(k5.values, k.values)
        if op == "NEO_HOOKEAN" and not per_point:
            mesh = _mesh(7)
            u = 0.002 * np.random.default_rng(8).standard_normal(3 * mesh.num_nodes())
            nodes = mesh.connectivity[4].astype(int)
            centre = mesh.vertices[nodes].mean(axis=0)
            for n in nodes:
                u[3 * n: 3 * n + 3] = -2.2 * (mesh.vertices[n] - centre)
            asm, ref = _build(eng, oracle, mesh, "NEO_HOOKEAN", u)
            k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
            vals = oracle.assemble(ref)[4]
            assert np.isnan(vals).any() and np.array_equal(np.isnan(k.values), np.isnan(vals))
            ok = ~np.isnan(vals)
            assert np.abs(k.values[ok] - vals[ok]).max() <= 1e-12 * np.abs(vals[ok]).max()
    finally:
        eng.close()


def test_large_deformation_keeps_the_tolerance(engine, oracle):
    """the trace term is formed as a_I^T (F F^T) a_J with a = F^-T g instead of g_I . g_J (hex27_blocks.hpp): its rounding error grows with
    cond(F)^2 -- a stretch of 3 x 1 x 1/3 with shear (cond F ~ 10) stays at 1e-11 of the largest entry"""
    mesh = _mesh(13, cells=(2, 2, 2))
    A = np.array([[2.0, 0.4, 0.0], [0.0, 0.0, 0.3], [0.2, 0.0, -2.0 / 3.0]])   # F = I + A
    u = (mesh.vertices @ A.T).reshape(-1)
    asm, ref = _build(engine, oracle, mesh, "NEO_HOOKEAN", u)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert engine.last_kernel_name() in HEX27_TWO_PASS
    vals = oracle.assemble(ref)[4]
    assert not np.isnan(vals).any()
    assert np.abs(k.values - vals).max() <= 1e-11 * np.abs(vals).max()


@pytest.mark.parametrize("k,layers", [(5, 2), (12, 2), (18, 2)])
def test_hex27_fans_through_the_triangle_gather(oracle, k, layers):
    """Hex27 elements around an axis (tests/test_high_valence.py's hexahedron fan, refined): the axis nodes belong to up to 2 k elements and their rows hold
    up to ~550 node blocks -- 16-bit column slots in the second pass (k >= 12), rows of very different lengths next to each other, several groups of
    entries per node -- against the oracle, symmetric bit for bit, the same bits twice, with few and with many nodes per wavefront."""
    from test_high_valence import hex_fan
    mesh = fa.hex27_mesh_from_hex8(hex_fan(k, layers))
    u = 0.01 * np.random.default_rng(k).standard_normal(3 * mesh.num_nodes())
    eng = fa.Engine(0)
    try:
        asm, ref = _build(eng, oracle, mesh, "NEO_HOOKEAN", u)
        k1 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert eng.last_kernel_name() in HEX27_TWO_PASS
        st, _, ro, ci, vals = oracle.assemble(ref)
        assert st == 0 and np.array_equal(k1.row_offsets, ro) and np.array_equal(k1.col_indices, ci)
        assert np.abs(k1.values - vals).max() <= 1e-12 * np.abs(vals).max()
        d = (k1.to_scipy() - k1.to_scipy().T).tocoo()
        assert d.nnz == 0 or not np.any(d.data != 0.0)
        for grid in (1, 7):
            eng.set_option("FENRIS_HIP_TWO_PASS_ROWS_GRID", grid)
            k2 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
            assert np.array_equal(k2.values, k1.values)
        eng.set_option("FENRIS_HIP_TWO_PASS_ROWS_GRID", None)
        eng.set_option("FENRIS_HIP_TWO_PASS_XCD_CHUNK", 8)
        eng.set_option("FENRIS_HIP_TWO_PASS_NODES_PER_WAVE", 1)
        k3 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert np.array_equal(k3.values, k1.values)
    finally:
        eng.close()
