"""Parity of the HIP engine (through the C ABI) against the CPU oracle on the same inputs.

Bar: row offsets / column indices bit-exact; values |K_gpu - K_oracle| <= 1e-12 max|K|;
NaN positions coincide for NeoHookean with J <= 0.
"""
import numpy as np
import pytest

import fenris_amd as fa

HEX27_TWO_PASS = ("k_hex27_dense_blocks + k_rows_from_tri",)
from fenris_amd import quadrature
from conftest import load_golden_mesh

pytestmark = pytest.mark.gpu

TOL = 1e-12
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2)) if True else None

OPS = {
    "LAPLACE": lambda: fa.LaplaceOperator(),
    "LINEAR_ELASTIC": lambda: fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
    "NEO_HOOKEAN": lambda: fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()),
    "STVK": lambda: fa.MaterialEllipticOperator(fa.StVKMaterial()),
}
KIND = {"QUAD4": fa.QUAD4, "HEX8": fa.HEX8, "TET4": fa.TET4, "HEX27": fa.HEX27, "TRI3": fa.TRI3}


@pytest.fixture(scope="module")
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


def _rule(kind):
    if kind == "QUAD4":
        return quadrature.tensor.quadrilateral_gauss(2)
    if kind == "HEX8":
        return quadrature.tensor.hexahedron_gauss(2)
    if kind == "HEX27":
        return quadrature.tensor.hexahedron_gauss(3)
    if kind == "TET4":
        return quadrature.total_order.tetrahedron(2)
    return quadrature.total_order.triangle(2)


def _mesh(kind, distort=True, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "QUAD4":
        m = fa.procedural.create_unit_square_uniform_quad_mesh_2d(5)
    elif kind == "HEX8":
        m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 4)
    elif kind == "HEX27":
        m = fa.hex27_mesh_from_hex8(fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 2))
    elif kind == "TET4":
        m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
    else:
        v, c = load_golden_mesh("square_quad4_79")
        tri = np.concatenate([c[:, [0, 1, 2]], c[:, [0, 2, 3]]])
        return fa.Mesh(v, tri, fa.TRI3)
    if distort:
        h = 1.0 / (5 if kind == "QUAD4" else 4)
        m = fa.Mesh(m.vertices + rng.uniform(-0.12 * h, 0.12 * h, m.vertices.shape), m.connectivity, m.elem_kind)
    return m


def _pair(engine, oracle, kind, op, mesh=None, u_scale=0.02, seed=1, params=None):
    """Build the GPU assembler and the oracle assembler on identical inputs."""
    mesh = mesh or _mesh(kind)
    w, p = _rule(kind)
    d = mesh.vertices.shape[1]
    s = 1 if op == "LAPLACE" else d
    rng = np.random.default_rng(seed)
    u = u_scale * rng.standard_normal(s * mesh.num_nodes())
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if op != "LAPLACE":
        if params is None:
            qt = qt.with_uniform_data(LAME)
            oparams = LAME.as_pair()
        else:
            qt = qt.with_data([fa.LameParameters(*x) for x in params])
            oparams = np.asarray(params)
    else:
        oparams = None
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(OPS[op]())
           .with_quadrature_table(qt).with_u(u).build())
    ref = oracle.ElementAssembler(KIND[kind], getattr(oracle, op), mesh.vertices, mesh.connectivity, w, p,
                                  params=oparams, u=u)
    return asm, ref


# ------------------------------------------------------------------------------------ pattern
MOCK = [[0, 1, 2], [2, 3], [], [3, 4, 4, 4, 4, 4, 4]]


def test_pattern_kats(engine):
    # tests/unit_tests/assembly/global.rs:70-142 (identical numbers in the parallel variant :144-216)
    ro, ci = fa.CsrAssembler().assemble_pattern(fa.MockElementAssembler(1, 0, [[]], engine))
    assert ro.tolist() == [0] and ci.tolist() == []
    ro, ci = fa.CsrAssembler().assemble_pattern(fa.MockElementAssembler(2, 5, [[]], engine))
    assert ro.tolist() == [0] * 11 and ci.tolist() == []
    ro, ci = fa.CsrAssembler().assemble_pattern(fa.MockElementAssembler(1, 6, MOCK, engine))
    assert ro.tolist() == [0, 3, 6, 10, 13, 15, 15]
    assert ci.tolist() == [0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 2, 3, 4, 3, 4]
    ro, ci = fa.CsrParAssembler().assemble_pattern(fa.MockElementAssembler(2, 6, MOCK, engine))
    assert ro.tolist() == [0, 6, 12, 18, 24, 32, 40, 46, 52, 56, 60, 60, 60]
    assert ci.tolist() == [
        0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 6, 7, 0, 1,
        2, 3, 4, 5, 6, 7, 4, 5, 6, 7, 8, 9, 4, 5, 6, 7, 8, 9, 6, 7, 8, 9, 6, 7, 8, 9]


def test_pattern_rejects_out_of_range_nodes(engine):
    with pytest.raises(fa.FenrisError):
        fa.MockElementAssembler(1, 3, [[0, 5]], engine)


@pytest.mark.parametrize("kind,op", [("HEX8", "LAPLACE"), ("HEX8", "LINEAR_ELASTIC"), ("TET4", "LINEAR_ELASTIC"),
                                     ("QUAD4", "LAPLACE"), ("QUAD4", "LINEAR_ELASTIC"), ("HEX27", "NEO_HOOKEAN"),
                                     ("TRI3", "LAPLACE")])
def test_pattern_matches_oracle_bit_exact(engine, oracle, kind, op):
    asm, ref = _pair(engine, oracle, kind, op)
    ro, ci = fa.CsrAssembler().assemble_pattern(asm)
    oro, oci = oracle.pattern_for(ref)
    assert ro.dtype == np.uint64 and np.array_equal(ro, oro) and np.array_equal(ci, oci)


def test_pattern_unstructured_fixtures(engine, oracle):
    # the reference's own unstructured meshes (tests/unit_tests/io/snapshots)
    for name, kind, op in (("sphere_tet4_593", "TET4", "LINEAR_ELASTIC"), ("cube_hex27_8", "HEX27", "LAPLACE"),
                           ("square_quad4_79", "QUAD4", "LAPLACE")):
        v, c = load_golden_mesh(name)
        asm, ref = _pair(engine, oracle, kind, op, mesh=fa.Mesh(v, c, KIND[kind]))
        ro, ci = asm.engine.pattern()
        oro, oci = oracle.pattern_for(ref)
        assert np.array_equal(ro, oro) and np.array_equal(ci, oci)


def test_pattern_device_arrays(engine, oracle):
    import torch

    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC")
    nnz = asm.engine.build_pattern()
    ro = torch.zeros(asm.engine.num_rows() + 1, dtype=torch.int64, device="cuda")
    ci = torch.zeros(nnz, dtype=torch.int64, device="cuda")
    asm.engine.pattern_dev(ro, ci)
    asm.engine.synchronize()
    oro, oci = oracle.pattern_for(ref)
    assert np.array_equal(ro.cpu().numpy().astype(np.uint64), oro)
    assert np.array_equal(ci.cpu().numpy().astype(np.uint64), oci)


# ------------------------------------------------------------------------------------ colouring
@pytest.mark.parametrize("kind", ["HEX8", "TET4", "QUAD4"])
def test_coloring_identical_to_reference_algorithm(engine, oracle, kind):
    asm, ref = _pair(engine, oracle, kind, "LAPLACE")
    colors = fa.color_nodes(asm)
    co, labels = oracle.color_nodes(ref)
    assert np.array_equal(colors.color_offsets, co) and np.array_equal(colors.labels, labels)
    if kind == "HEX8":
        assert len(colors) == 8


@pytest.mark.parametrize("kind", ["HEX8", "TET4", "QUAD4", "HEX27"])
def test_parallel_coloring_is_a_valid_deterministic_coloring(engine, oracle, kind):
    """fh_color_parallel: the colouring computed on the device -- every element once, ascending inside a colour, no two elements of a
    colour sharing a node (the precondition of DisjointSubsets, fenris-paradis/src/lib.rs), the same result every time; and the
    coloured scatter driven by it gives the oracle's matrix"""
    asm, ref = _pair(engine, oracle, kind, "LAPLACE")
    colors = asm.engine.color_parallel()
    E = asm.num_elements()
    assert sorted(colors.labels.tolist()) == list(range(E))
    conn = np.asarray(asm.space.connectivity).astype(np.int64)
    for c in range(len(colors)):
        lab = colors.color(c).astype(np.int64)
        assert np.all(np.diff(lab) > 0)
        nodes = conn[lab].ravel()
        assert len(np.unique(nodes)) == len(nodes)
    again = asm.engine.color_parallel()
    assert np.array_equal(again.color_offsets, colors.color_offsets) and np.array_equal(again.labels, colors.labels)
    seq = fa.color_nodes(asm)
    assert len(colors) >= len(seq) or len(colors) >= 1   # the sequential greedy colouring is not a lower bound, only a yardstick
    k = fa.CsrParAssembler().assemble(colors, asm)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert np.array_equal(k.col_indices, oci)
    assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max()


def test_coloring_ragged(engine, oracle):
    mock = fa.MockElementAssembler(1, 6, MOCK, engine)
    colors = mock.engine.color()
    offs = np.cumsum([0] + [len(c) for c in MOCK]).astype(np.uint64)
    nodes = np.array([x for c in MOCK for x in c], dtype=np.uint64)
    co, labels = oracle.color_elements(offs, nodes)
    assert np.array_equal(colors.color_offsets, co) and np.array_equal(colors.labels, labels)


# ------------------------------------------------------------------------------------ element matrices
@pytest.mark.parametrize("kind", ["QUAD4", "HEX8", "TET4", "HEX27", "TRI3"])
@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC", "NEO_HOOKEAN", "STVK"])
def test_element_matrices_match_oracle(engine, oracle, kind, op):
    asm, ref = _pair(engine, oracle, kind, op)
    E = asm.num_elements()
    ke = asm.engine.element_matrices(0, E)
    for e in range(E):
        st, oke = ref.element_matrix(e)
        assert st == 0
        nan = np.isnan(oke)  # NeoHookean with J <= 0: NaN positions must coincide (materials.rs:298-300)
        assert np.array_equal(np.isnan(ke[e]), nan)
        if nan.all():
            continue
        scale = np.abs(oke[~nan]).max()
        assert np.abs(ke[e][~nan] - oke[~nan]).max() <= TOL * scale, (e, np.abs(ke[e][~nan] - oke[~nan]).max() / scale)
        assert np.array_equal(ke[e][~nan], ke[e].T[~nan])


def test_quad4_reference_element_kat(engine):
    # tests/unit_tests/assembly.rs:159-162
    mesh = fa.Mesh(np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=float), np.array([[0, 1, 2, 3]]), fa.QUAD4)
    w, p = quadrature.tensor.quadrilateral_gauss(2)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(fa.LaplaceOperator())
           .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w)).with_u(np.zeros(4)).build())
    expected = np.array([[2 / 3, -1 / 6, -1 / 3, -1 / 6], [-1 / 6, 2 / 3, -1 / 6, -1 / 3],
                         [-1 / 3, -1 / 6, 2 / 3, -1 / 6], [-1 / 6, -1 / 3, -1 / 6, 2 / 3]])
    assert np.allclose(asm.assemble_element_matrix(0), expected, rtol=0, atol=1e-15)


def test_per_point_parameters(engine, oracle):
    params = [(1e6 * (1 + 0.1 * q), 2e5 * (1 + 0.05 * q)) for q in range(8)]
    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC", params=params)
    ke = asm.engine.element_matrices(0, 4)
    for e in range(4):
        _, oke = ref.element_matrix(e)
        assert np.abs(ke[e] - oke).max() <= TOL * np.abs(oke).max()


# ------------------------------------------------------------------------------------ global matrix
@pytest.mark.parametrize("scatter", [fa.SCATTER_ATOMIC, fa.SCATTER_COLORED, fa.SCATTER_GATHER])
@pytest.mark.parametrize("kind,op", [("HEX8", "LAPLACE"), ("HEX8", "LINEAR_ELASTIC"), ("HEX8", "NEO_HOOKEAN"),
                                     ("TET4", "LAPLACE"), ("TET4", "LINEAR_ELASTIC"), ("TET4", "STVK"),
                                     ("QUAD4", "LAPLACE"), ("QUAD4", "LINEAR_ELASTIC"),
                                     ("HEX27", "LAPLACE"), ("HEX27", "NEO_HOOKEAN"), ("TRI3", "LINEAR_ELASTIC")])
def test_global_matrix_matches_oracle(engine, oracle, kind, op, scatter):
    asm, ref = _pair(engine, oracle, kind, op)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert st == 0
    k = fa.CsrAssembler(scatter).assemble(asm)
    assert np.array_equal(k.row_offsets, oro) and np.array_equal(k.col_indices, oci)
    err = np.abs(k.values - ovals).max() / np.abs(ovals).max()
    assert err <= TOL, err


def test_unstructured_sphere_tet4(engine, oracle):
    v, c = load_golden_mesh("sphere_tet4_593")
    mesh = fa.Mesh(v, c, fa.TET4)
    for scatter in (fa.SCATTER_ATOMIC, fa.SCATTER_COLORED, fa.SCATTER_GATHER):
        asm, ref = _pair(engine, oracle, "TET4", "LINEAR_ELASTIC", mesh=mesh)
        st, _, oro, oci, ovals = oracle.assemble(ref)
        k = fa.CsrAssembler(scatter).assemble(asm)
        assert np.array_equal(k.col_indices, oci)
        assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max()


def test_par_assembler_with_reference_colors(engine, oracle):
    # tests/convergence_tests/poisson_mms_common.rs:115-121: serial == coloured
    asm, ref = _pair(engine, oracle, "HEX8", "LAPLACE")
    colors = fa.color_nodes(asm)
    a = fa.CsrAssembler().assemble(asm)
    b = fa.CsrParAssembler().assemble(colors, asm)
    assert np.abs(a.values - b.values).max() <= 8 * np.finfo(float).eps * np.abs(a.values).max()


@pytest.mark.parametrize("scatter", [fa.SCATTER_ATOMIC, fa.SCATTER_COLORED, fa.SCATTER_GATHER])
def test_assemble_into_accumulates_and_overwrite(engine, oracle, scatter):
    # global.rs:133-182: assemble_into_csr adds to the existing values
    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC")
    k = fa.CsrAssembler(scatter).assemble(asm)
    base = k.values.copy()
    fa.CsrAssembler(scatter).assemble_into_csr(k, asm)
    assert np.abs(k.values - 2 * base).max() <= 4e-16 * np.abs(base).max() * 4
    k.values[:] = 123.0
    if scatter == fa.SCATTER_COLORED:
        asm.engine.color()
    asm.engine.assemble_matrix(k.values, scatter | fa.ASSEMBLE_OVERWRITE)
    assert np.abs(k.values - base).max() <= 4e-16 * np.abs(base).max() * 4


def test_device_resident_values(engine, oracle):
    import torch

    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC")
    st, _, oro, oci, ovals = oracle.assemble(ref)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm, device_values=True)
    assert isinstance(k.values, torch.Tensor) and k.values.is_cuda
    assert np.abs(k.values.cpu().numpy() - ovals).max() <= TOL * np.abs(ovals).max()


def test_repeated_runs_agree(engine, oracle):
    # race-detection analogue of fenris-paradis/test_sanitized.sh: repeated runs and all three scatter
    # strategies agree to rounding
    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC")
    runs = [fa.CsrAssembler(s).assemble(asm).values for s in (0, 0, 1, 2, 2)]
    scale = np.abs(runs[0]).max()
    for r in runs[1:]:
        assert np.abs(r - runs[0]).max() <= 16 * np.finfo(float).eps * scale


# ------------------------------------------------------------------------------------ errors / NaN semantics
def test_singular_jacobian_reports_lowest_element(engine):
    mesh = _mesh("HEX8", distort=False)
    v = mesh.vertices.copy()
    c = mesh.connectivity
    for e in (17, 5):  # collapse two elements to a point -> det == 0 exactly
        v[c[e].astype(int)] = v[int(c[e][0])]
    bad = fa.Mesh(v, c, fa.HEX8)
    w, p = _rule("HEX8")
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(bad).with_operator(fa.LaplaceOperator())
           .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w)).with_u(np.zeros(len(v))).build())
    for scatter in (fa.SCATTER_ATOMIC, fa.SCATTER_GATHER):
        with pytest.raises(fa.SingularJacobianError) as ei:
            fa.CsrAssembler(scatter).assemble(asm)
        assert "Singular element Jacobian encountered" in str(ei.value)
        assert ei.value.element <= 5


def test_neo_hookean_inverted_element_nan_positions(engine, oracle):
    # fenris-solid/src/materials.rs:298-300: J <= 0 => all-NaN blocks, not an error
    mesh = _mesh("HEX8", distort=False)
    rng = np.random.default_rng(3)
    u = 0.01 * rng.standard_normal(3 * mesh.num_nodes())
    nodes = mesh.connectivity[10].astype(int)
    centre = mesh.vertices[nodes].mean(axis=0)
    for n in nodes:  # reflect element 10 through its centre: F = -I there, det F < 0
        u[3 * n: 3 * n + 3] = -2.2 * (mesh.vertices[n] - centre)
    w, p = _rule("HEX8")
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh)
           .with_operator(OPS["NEO_HOOKEAN"]()).with_quadrature_table(qt).with_u(u).build())
    ref = oracle.ElementAssembler(oracle.HEX8, oracle.NEO_HOOKEAN, mesh.vertices, mesh.connectivity, w, p,
                                  params=LAME.as_pair(), u=u)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert st == 0 and np.isnan(ovals).any() and not np.isnan(ovals).all()
    for scatter in (fa.SCATTER_ATOMIC, fa.SCATTER_GATHER):
        k = fa.CsrAssembler(scatter).assemble(asm)
        assert np.array_equal(np.isnan(k.values), np.isnan(ovals))
        ok = ~np.isnan(ovals)
        assert np.abs(k.values[ok] - ovals[ok]).max() <= TOL * np.abs(ovals[ok]).max()


# ------------------------------------------------------------------------------------ vector / scalar
@pytest.mark.parametrize("kind,op", [("HEX8", "LAPLACE"), ("HEX8", "LINEAR_ELASTIC"), ("HEX8", "NEO_HOOKEAN"),
                                     ("HEX8", "STVK"), ("TET4", "NEO_HOOKEAN"), ("QUAD4", "LINEAR_ELASTIC"),
                                     ("HEX27", "NEO_HOOKEAN"), ("TRI3", "LAPLACE")])
def test_vector_and_scalar_match_oracle(engine, oracle, kind, op):
    asm, ref = _pair(engine, oracle, kind, op)
    f = fa.VectorAssembler().assemble_vector(asm)
    st, _, of = oracle.assemble_vector(ref)
    assert st == 0
    assert np.abs(f - of).max() <= TOL * np.abs(of).max()
    e = fa.assemble_scalar(asm)
    st, _, oe = oracle.assemble_scalar(ref)
    assert e == pytest.approx(oe, rel=1e-12)
    # accumulation semantics (global.rs:582-608)
    out = f.copy()
    fa.VectorAssembler().assemble_vector_into(out, asm)
    assert np.abs(out - 2 * f).max() <= 1e-15 * np.abs(f).max() * 8
    if kind in ("HEX8", "TET4", "QUAD4"):
        # two-pass residual (element vectors, then a per-row sum in ascending element order): no atomics, same bits every run
        assert np.array_equal(fa.VectorAssembler().assemble_vector(asm), f)


def test_vector_async_matches_the_blocking_call_and_reports_through_poll(engine, oracle):
    """fh_assemble_vector_async_dev only enqueues: same bits as fh_assemble_vector_dev, a singular element comes out of fh_poll_status"""
    import torch

    asm, ref = _pair(engine, oracle, "HEX8", "NEO_HOOKEAN")
    n = asm.solution_dim() * asm.num_nodes()
    a = torch.zeros(n, dtype=torch.float64, device="cuda")
    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    engine.assemble_vector(a)
    engine.assemble_vector_async(b)
    engine.poll_status()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    st, _, of = oracle.assemble_vector(ref)
    assert np.abs(a.cpu().numpy() - of).max() <= TOL * np.abs(of).max()
    # a collapsed element: the enqueue succeeds, the poll raises
    mesh = _mesh("HEX8", distort=False)
    v = mesh.vertices.copy()
    v[mesh.connectivity[7].astype(int)] = v[int(mesh.connectivity[7][0])]
    w, p = _rule("HEX8")
    bad = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(fa.Mesh(v, mesh.connectivity, fa.HEX8))
           .with_operator(fa.LaplaceOperator()).with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w))
           .with_u(np.zeros(len(v))).build())
    assert bad.engine is engine
    c = torch.zeros(len(v), dtype=torch.float64, device="cuda")
    engine.assemble_vector_async(c)
    with pytest.raises(fa.SingularJacobianError):
        engine.poll_status()


@pytest.mark.parametrize("kind,op", [("TET4", "LINEAR_ELASTIC"), ("TET4", "LAPLACE"), ("HEX8", "LINEAR_ELASTIC"), ("QUAD4", "LAPLACE"),
                                     ("TRI3", "LINEAR_ELASTIC"), ("HEX8", "NEO_HOOKEAN"), ("TET4", "STVK"), ("HEX27", "LINEAR_ELASTIC"),
                                     ("HEX27", "NEO_HOOKEAN")])
def test_overwrite_with_a_mask_leaves_nothing_stale(engine, oracle, kind, op):
    """FH_ASSEMBLE_OVERWRITE into an array full of garbage, with an element mask: blocks of the pattern that no ACTIVE element touches must
    come out as zeros (the owner-computes kernels write every value of their rows exactly once and have no zero-fill pass)"""
    import torch

    asm, ref = _pair(engine, oracle, kind, op)
    eng = asm.engine
    nnz = eng.build_pattern()
    E = asm.num_elements()
    active = (np.arange(E) % 3 != 1)
    eng.set_active_elements(active)
    try:
        want = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(want, fa.SCATTER_ATOMIC)
        got = torch.full((nnz,), 7.25, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(got, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        w, g = want.cpu().numpy(), got.cpu().numpy()
        assert np.abs(g - w).max() <= TOL * np.abs(w).max(), eng.last_kernel_name()
    finally:
        eng.set_active_elements(None)
    # and without a mask
    want = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    eng.assemble_matrix(want, fa.SCATTER_ATOMIC)
    got = torch.full((nnz,), -3.5, dtype=torch.float64, device="cuda")
    eng.assemble_matrix(got, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    w, g = want.cpu().numpy(), got.cpu().numpy()
    assert np.abs(g - w).max() <= TOL * np.abs(w).max(), eng.last_kernel_name()


@pytest.mark.parametrize("kind,op", [("HEX8", "LINEAR_ELASTIC"), ("TET4", "LAPLACE"), ("HEX27", "NEO_HOOKEAN"), ("QUAD4", "STVK")])
def test_mask_without_an_active_element(engine, oracle, kind, op):
    """an element mask that switches EVERYTHING off (a rank whose slab is all halo, a selection that came out empty): every scatter gives
    the zero matrix / leaves an accumulated one alone, vector and energy are zero -- and no launch of zero workgroups reaches the runtime"""
    import torch

    asm, ref = _pair(engine, oracle, kind, op)
    eng = asm.engine
    nnz = eng.build_pattern()
    eng.set_active_elements(np.zeros(asm.num_elements(), dtype=np.uint8))
    try:
        for scatter in (fa.SCATTER_ATOMIC, fa.SCATTER_GATHER, fa.SCATTER_COLORED):
            if scatter == fa.SCATTER_COLORED:
                eng.color()
            v = torch.full((nnz,), 3.25, dtype=torch.float64, device="cuda")
            eng.assemble_matrix(v, scatter)                                   # accumulate: untouched
            assert torch.all(v == 3.25)
            eng.assemble_matrix(v, scatter | fa.ASSEMBLE_OVERWRITE)           # overwrite: zeros
            assert torch.all(v == 0.0), eng.last_kernel_name()
        f = torch.full((asm.solution_dim() * asm.num_nodes(),), 1.5, dtype=torch.float64, device="cuda")
        eng.assemble_vector(f)
        assert torch.all(f == 1.5)
    finally:
        eng.set_active_elements(None)


def test_residual_is_K_times_u_for_linear_operators(engine, oracle):
    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC", u_scale=1e-3)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm).to_scipy()
    f = fa.VectorAssembler().assemble_vector(asm)
    assert np.abs(f - k @ ref.u).max() <= 1e-11 * np.abs(f).max()


# ------------------------------------------------------------------------------------ Dirichlet helper
def test_dirichlet_on_device_matches_oracle(engine, oracle):
    import torch

    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC")
    st, _, oro, oci, ovals = oracle.assemble(ref)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm, device_values=True)
    bc = np.where(np.abs(asm.space.vertices - 0.5).max(axis=1) > 0.4)[0]
    fa.apply_homogeneous_dirichlet_bc_csr(k, bc, 3, asm)
    ov = ovals.copy()
    oracle.apply_homogeneous_dirichlet_bc_csr(oro, oci, ov, bc, 3)
    got = k.values.cpu().numpy()
    assert np.abs(got - ov).max() <= TOL * np.abs(ov).max()
    assert np.array_equal(got == 0.0, ov == 0.0)


# ------------------------------------------------------------------------------------ full size properties
@pytest.mark.parametrize("cells,op", [(128, "LAPLACE"), (216, "LINEAR_ELASTIC")])
def test_full_size_properties(engine, oracle, cells, op):
    """BASELINE.json configs C2 (Hex8 128^3 Poisson) and NS (Hex8 216^3 elasticity) at full size, checked
    through size-independent properties: (1) constant fields are in the null space (row sums vanish per
    component), (2) owner-computes and atomic scatter agree, (3) an interior row block equals the centre
    row block of a 3^3 oracle mesh with the same cell size (translation invariance), (4) pattern counts
    match the closed forms of SURVEY 8 (nnz = s^2 (3m+1)^3)."""
    import torch

    mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(cells)
    w, p = _rule("HEX8")
    s = 1 if op == "LAPLACE" else 3
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if s == 3:
        qt = qt.with_uniform_data(LAME)
    eng = fa.Engine(0)
    asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(OPS[op]())
           .with_quadrature_table(qt).with_u(None).build())
    nnz = eng.build_pattern()
    assert nnz == s * s * (3 * cells + 1) ** 3
    ro, _ = eng.pattern(want_cols=False)
    vals = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    eng.assemble_matrix(vals, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    scale = float(vals.abs().max())
    # (1) row sums per column component
    ro_t = torch.from_numpy(ro.astype(np.int64)).cuda()
    R = len(ro) - 1
    lengths = ro_t[1:] - ro_t[:-1]
    row_id = torch.repeat_interleave(torch.arange(R, device="cuda"), lengths)
    local = torch.arange(nnz, device="cuda") - ro_t[row_id]
    for comp in range(s):
        sums = torch.zeros(R, dtype=torch.float64, device="cuda").index_add_(0, row_id, vals * (local % s == comp))
        # a sum of 27 terms of size `scale` that cancel: TOL per term
        assert float(sums.abs().max()) <= 27 * TOL * scale, float(sums.abs().max()) / scale
    del row_id, local, lengths
    # (2) atomic scatter on the same pattern
    vals2 = torch.zeros_like(vals)
    eng.assemble_matrix(vals2, fa.SCATTER_ATOMIC)
    assert float((vals - vals2).abs().max()) <= 1e-13 * scale
    del vals2
    # (3) interior row block against the oracle on a 3x3x3 mesh of equal cell size
    h = 1.0 / cells
    small = fa.procedural.create_rectangular_uniform_hex_mesh(3 * h, 1, 1, 1, 3)
    ref = oracle.ElementAssembler(oracle.HEX8, getattr(oracle, op), small.vertices, small.connectivity, w, p,
                                  params=None if s == 1 else LAME.as_pair())
    st, _, oro, oci, ovals = oracle.assemble(ref)
    centre = 1 + 4 * 1 + 16 * 1
    oblock = ovals[int(oro[s * centre]): int(oro[s * centre + s - 1 + 1])]
    nv = cells + 1
    node = (cells // 2) + nv * (cells // 3) + nv * nv * (cells // 2 + 3)
    block = vals[int(ro[s * node]): int(ro[s * node + s])].cpu().numpy()
    assert block.shape == oblock.shape
    assert np.abs(block - oblock).max() <= TOL * np.abs(oblock).max(), np.abs(block - oblock).max() / np.abs(oblock).max()
    eng.close()


# ------------------------------------------------------------------------------------ end-to-end golden (MMS)
@pytest.mark.parametrize("name,kind,nres", [("poisson2d_mms_quad4_summary", "QUAD4", 5),
                                            ("poisson3d_mms_hex8_summary", "HEX8", 4),
                                            ("poisson3d_mms_tet4_summary", "TET4", 3)])
def test_mms_errors_with_gpu_assembled_matrix(engine, oracle, name, kind, nres):
    """tests/convergence_tests/poisson_{2d,3d}_mms.rs + reference_values/*.json (tolerance 1 %,
    poisson_mms_common.rs:40-65): the stiffness matrix comes from the HIP engine (owner-computes kernel),
    Dirichlet rows/columns are applied on the device, the source vector / solve / error norms are the
    numpy/scipy restatement used to pin the oracle."""
    import json
    import os

    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch

    from conftest import GOLDEN

    ref = json.load(open(os.path.join(GOLDEN, "mms_reference_values.json")))["summaries"][name]
    okind = getattr(oracle, kind)
    if kind == "QUAD4":
        gen, rule, err_rule = fa.procedural.create_unit_square_uniform_quad_mesh_2d, quadrature.tensor.quadrilateral_gauss(2), oracle.quadrilateral_gauss(6)
    elif kind == "HEX8":
        gen, rule, err_rule = fa.procedural.create_unit_box_uniform_hex_mesh_3d, quadrature.tensor.hexahedron_gauss(2), oracle.hexahedron_gauss(6)
    else:
        t = json.load(open(os.path.join(GOLDEN, "tet_rule_6_24.json")))
        gen, rule, err_rule = (fa.procedural.create_unit_box_uniform_tet_mesh_3d, quadrature.total_order.tetrahedron(0),
                               (np.array(t["weights"]), np.array(t["points"])))
    for i, res in enumerate([1, 2, 4, 8, 16][:nres]):
        mesh = gen(res)
        w, p = rule
        v, c = mesh.vertices, mesh.connectivity
        d = v.shape[1]
        N = len(v)
        asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(fa.LaplaceOperator())
               .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w)).with_u(np.zeros(N)).build())
        k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm, device_values=True)
        bc = np.where(np.abs(v - 0.5).max(axis=1) > 0.4999)[0]
        fa.apply_homogeneous_dirichlet_bc_csr(k, bc, 1, asm)
        A = k.to_scipy()
        u_exact = lambda x: np.prod(np.sin(np.pi * x), axis=-1)
        b = np.zeros(N)
        ci = c.astype(int)
        for e in range(len(c)):
            X = v[ci[e]]
            for wq, xi in zip(w, p):
                J = X.T @ oracle.element_gradients(okind, xi).T
                x = oracle.element_basis(okind, xi) @ X
                b[ci[e]] += wq * abs(np.linalg.det(J)) * (d * np.pi ** 2 * u_exact(x)) * oracle.element_basis(okind, xi)
        b[bc] = 0.0
        u_h = spla.spsolve(A.tocsc(), b)
        l2 = 0.0
        for e in range(len(c)):
            X = v[ci[e]]
            for wq, xi in zip(*err_rule):
                J = X.T @ oracle.element_gradients(okind, xi).T
                x = oracle.element_basis(okind, xi) @ X
                l2 += wq * abs(np.linalg.det(J)) * (oracle.element_basis(okind, xi) @ u_h[ci[e]] - u_exact(x)) ** 2
        l2 = np.sqrt(l2)
        assert abs(l2 - ref["L2_errors"][i]) / ref["L2_errors"][i] < 0.01, (res, l2, ref["L2_errors"][i])


# ------------------------------------------------------------------------------------ edge cases
def test_isolated_vertices_and_empty_mesh(engine, oracle):
    """Vertices that belong to no element give empty CSR rows (like the reference: node_sets stay empty,
    global.rs:69-93); a mesh without elements assembles to nothing."""
    base = _mesh("HEX8", distort=True)
    extra = np.array([[5.0, 5.0, 5.0], [6.0, 5.0, 5.0]])
    v = np.vstack([base.vertices[:10], extra, base.vertices[10:]])
    remap = np.concatenate([np.arange(10), np.arange(10, len(base.vertices)) + 2]).astype(np.uint64)
    mesh = fa.Mesh(v, remap[base.connectivity.astype(int)], fa.HEX8)
    for scatter in (fa.SCATTER_ATOMIC, fa.SCATTER_COLORED, fa.SCATTER_GATHER):
        asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC", mesh=mesh)
        st, _, oro, oci, ovals = oracle.assemble(ref)
        assert st == 0
        k = fa.CsrAssembler(scatter).assemble(asm)
        assert np.array_equal(k.row_offsets, oro) and np.array_equal(k.col_indices, oci)
        assert oro[3 * 10] == oro[3 * 12]  # the two isolated vertices own empty rows
        assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max()
    empty = fa.Mesh(base.vertices, np.zeros((0, 8), dtype=np.uint64), fa.HEX8)
    w, p = _rule("HEX8")
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(empty).with_operator(fa.LaplaceOperator())
           .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w)).with_u(None).build())
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert len(k.values) == 0 and k.row_offsets.tolist() == [0] * (len(base.vertices) + 1)
    assert fa.VectorAssembler().assemble_vector(asm).tolist() == [0.0] * len(base.vertices)
    assert fa.assemble_scalar(asm) == 0.0


def test_benchmark_kernels_are_the_ones_the_parity_tests_cover(engine, oracle):
    """bench.py's kernels -- k_affine_rows (structured boxes: the headline) and k_hex8_rows (every other Hex8 mesh with the eight-point
    rule; round 4, before: k_gather_pipelined) -- must be the code paths that the parity tests compare with the oracle."""
    mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(12)
    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC", mesh=mesh)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    # a structured box consists of affine elements: the affine-element form of the owner-computes kernel (bench.py's headline)
    assert asm.engine.last_kernel_name() == "k_affine_rows"
    assert np.array_equal(k.col_indices, oci)
    assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max()
    # the same mesh through the general kernel (affine detection switched off), the one perturbed meshes take
    asm.engine.set_affine_tolerance(0.0)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert asm.engine.last_kernel_name() == "k_hex8_rows"
    assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max()
    asm.engine.set_affine_tolerance(2.0 ** -46)
    # distorted (non-affine) elements and the Tet4 / Quad4 instantiations of the same kernel
    for kind in ("HEX8", "TET4", "QUAD4", "TRI3"):
        for op in ("LAPLACE", "LINEAR_ELASTIC"):
            asm, ref = _pair(engine, oracle, kind, op)
            st, _, oro, oci, ovals = oracle.assemble(ref)
            k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
            # Tet4 takes the row-owner form of the kernel (rows_kernel.hpp); its four-point rule is collapsed to one point.  Hex8 with
            # the eight-point rule: row-owner lanes as well (hex8_rows.hip); Quad4 / Tri3: the pipelined kernel
            want = {"TET4": "k_gather_rows", "HEX8": "k_hex8_rows"}.get(kind, "k_gather_pipelined")
            assert asm.engine.last_kernel_name() == want, (kind, op)
            assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max(), (kind, op)


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
def test_pipelined_kernel_mirrored_and_permuted_hex8(engine, oracle, op):
    """The planar instantiation folds sign(det J) sqrt(w) / sqrt(|det J|) into the adjugate: elements with a negative
    Jacobian determinant (mirrored node order), a random vertex / element numbering and widely varying element
    sizes must come out like the oracle's."""
    base = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 2, 1, 1, 5)
    rng = np.random.default_rng(7)
    v = base.vertices.copy()
    v += 0.03 * rng.standard_normal(v.shape)
    v[:, 0] = np.sign(v[:, 0]) * np.abs(v[:, 0]) ** 2.5      # graded: element sizes over two orders of magnitude
    conn = base.connectivity.copy()
    flip = rng.random(len(conn)) < 0.5
    conn[flip] = conn[flip][:, [4, 5, 6, 7, 0, 1, 2, 3]]     # swap bottom and top face: det J < 0
    perm = rng.permutation(len(v))
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(v))
    mesh = fa.Mesh(v[perm], inv[conn][rng.permutation(len(conn))], fa.HEX8)
    asm, ref = _pair(engine, oracle, "HEX8", op, mesh=mesh)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert st == 0
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert asm.engine.last_kernel_name() == "k_gather_pipelined"
    assert np.array_equal(k.row_offsets, oro) and np.array_equal(k.col_indices, oci)
    assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max()
    ka = fa.CsrAssembler(fa.SCATTER_ATOMIC).assemble(asm)
    assert np.abs(k.values - ka.values).max() <= TOL * np.abs(ovals).max()


class _Shifted:
    """a slice of a long array addressed with the indices of the whole"""

    def __init__(self, part, first):
        self.part, self.first = part, first

    def __getitem__(self, sl):
        return self.part[sl.start - self.first: sl.stop - self.first]


def test_full_size_tet4_elasticity_properties(engine, oracle):
    """BASELINE config C3 (Tet4 linear elasticity, BCC res 75, vertices and elements permuted) at full size:
    closed-form nnz, owner-computes == atomic == coloured, rigid translations in the null space, and the row
    block of one interior lattice node equals the oracle's on a small BCC mesh with the same cell size."""
    import torch

    res = 75
    m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(res)
    rng = np.random.Generator(np.random.MT19937(12345))
    vp = rng.permutation(m.num_nodes())
    inv = np.empty_like(vp)
    inv[vp] = np.arange(len(vp))
    mesh = fa.Mesh(m.vertices[vp], inv[m.connectivity.astype(np.int64)][rng.permutation(m.num_elements())].astype(np.uint64), fa.TET4)
    w, p = quadrature.total_order.tetrahedron(1)
    eng = fa.Engine(0)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)
    (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(OPS["LINEAR_ELASTIC"]())
     .with_quadrature_table(qt).with_u(None).build())
    nnz = eng.build_pattern()
    assert nnz == 9 * (30 * res ** 3 + 21 * res ** 2 + 9 * res + 1)
    ro, ci = eng.pattern()
    vals = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    eng.assemble_matrix(vals, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    scale = float(vals.abs().max())
    for scatter in (fa.SCATTER_ATOMIC, fa.SCATTER_COLORED):
        v2 = torch.zeros_like(vals)
        if scatter == fa.SCATTER_COLORED:
            eng.color()
        eng.assemble_matrix(v2, scatter)
        assert float((vals - v2).abs().max()) <= 1e-12 * scale
    ro_t = torch.from_numpy(ro.astype(np.int64)).cuda()
    R = len(ro) - 1
    row_id = torch.repeat_interleave(torch.arange(R, device="cuda"), ro_t[1:] - ro_t[:-1])
    local = torch.arange(nnz, device="cuda") - ro_t[row_id]
    for comp in range(3):
        sums = torch.zeros(R, dtype=torch.float64, device="cuda").index_add_(0, row_id, vals * (local % 3 == comp))
        assert float(sums.abs().max()) <= 15 * TOL * scale, float(sums.abs().max()) / scale   # 15 cancelling terms per row and component
    # interior lattice node (37,40,33) of the generator numbering -> permuted index; compare its three rows, column node by
    # column node, with the rows of the centre lattice node of a 4^3-cell oracle mesh of equal cell size
    h = 1.0 / res
    small = fa.procedural.create_rectangular_uniform_tet_mesh(4 * h, 1, 1, 1, 4)
    ref = oracle.ElementAssembler(oracle.TET4, oracle.LINEAR_ELASTIC, small.vertices, small.connectivity, w, p, params=LAME.as_pair())
    st, _, oro, oci, ovals = oracle.assemble(ref)
    c = 2 + 5 * 2 + 25 * 2
    node = int(inv[37 + 76 * 40 + 76 * 76 * 33])

    def rows_by_neighbour(ro_, ci_, values_, verts, nd):
        """{offset of the column node from `nd` in half cells: the 3 x 3 block (row component, column component)}"""
        out = {}
        for r in range(3):
            lo, hi = int(ro_[3 * nd + r]), int(ro_[3 * nd + r + 1])
            cols = ci_[lo:hi].astype(np.int64)
            for k in range(0, hi - lo, 3):
                j = int(cols[k]) // 3
                assert cols[k] == 3 * j and cols[k + 1] == 3 * j + 1 and cols[k + 2] == 3 * j + 2
                key = tuple(np.rint(2.0 * (verts[j] - verts[nd]) / h).astype(int))
                out.setdefault(key, np.zeros((3, 3)))[r] = values_[lo + k: lo + k + 3]
        return out

    lo, hi = int(ro[3 * node]), int(ro[3 * node + 3])
    got = rows_by_neighbour(ro, ci, _Shifted(vals[lo:hi].cpu().numpy(), lo), mesh.vertices, node)
    want = rows_by_neighbour(oro, oci, ovals, small.vertices, c)
    assert set(got) == set(want) and len(got) == 15  # same neighbours (the permutation changes their order, not who they are)
    err = max(np.abs(got[k] - want[k]).max() for k in want)
    assert err <= TOL * np.abs(ovals).max(), err / np.abs(ovals).max()
    eng.close()


def test_full_size_hex27_neo_hookean_properties(engine, oracle):
    """BASELINE config C4 (Hex27 NeoHookean 50x50x80, 27-point rule) at full size: closed-form nnz, atomic and
    coloured scatter agree, values finite for the homogeneous deformation u = 0.05 A X."""
    import torch

    h8 = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 5, 5, 8, 10)
    mesh = fa.hex27_mesh_from_hex8(h8)
    assert mesh.num_nodes() == 101 * 101 * 161
    A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
    u = (0.05 * mesh.vertices @ A.T).reshape(-1)
    w, p = quadrature.tensor.hexahedron_gauss(3)
    eng = fa.Engine(0)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)
    (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(OPS["NEO_HOOKEAN"]())
     .with_quadrature_table(qt).with_u(u).build())
    nnz = eng.build_pattern()
    assert nnz == 9 * 401 * 401 * 641  # 9 * (8m+1) per direction
    vals = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    eng.color()
    eng.assemble_matrix(vals, fa.SCATTER_COLORED | fa.ASSEMBLE_OVERWRITE)
    assert bool(torch.isfinite(vals).all())
    v2 = torch.zeros_like(vals)
    eng.assemble_matrix(v2, fa.SCATTER_ATOMIC)
    assert float((vals - v2).abs().max()) <= 1e-12 * float(vals.abs().max())
    # the production path for high-order elements: two-pass owner-computes, element matrices on the fp64 matrix cores
    v3 = torch.empty_like(vals).fill_(float("nan"))
    eng.assemble_matrix(v3, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    assert eng.last_kernel_name() in HEX27_TWO_PASS
    assert float((vals - v3).abs().max()) <= 1e-12 * float(vals.abs().max())
    # size-independent properties through the blocked-CSR SpMV: a rigid translation carries no force (every elastic
    # tangent annihilates constant displacement fields), and K is symmetric: x.(K y) == y.(K x)
    n = 3 * mesh.num_nodes()
    scale = float(vals.abs().max())
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    for comp in range(3):
        t = torch.zeros(n, dtype=torch.float64, device="cuda")
        t[comp::3] = 1.0
        eng.spmv(v3, t, y)
        assert float(y.abs().max()) <= 1e-10 * scale
    g = torch.Generator(device="cuda").manual_seed(7)
    x1 = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    x2 = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    eng.spmv(v3, x2, y)
    a12 = float(torch.dot(x1, y))
    eng.spmv(v3, x1, y)
    a21 = float(torch.dot(x2, y))
    assert abs(a12 - a21) <= 1e-10 * abs(a12)
    del v3, x1, x2, y
    # one element matrix against the oracle (same inputs)
    ref = oracle.ElementAssembler(oracle.HEX27, oracle.NEO_HOOKEAN, mesh.vertices, mesh.connectivity[:3], w, p,
                                  params=LAME.as_pair(), u=u)
    ke = eng.element_matrices(1, 1)[0]
    st, oke = ref.element_matrix(1)
    assert st == 0 and np.abs(ke - oke).max() <= TOL * np.abs(oke).max()
    eng.close()


def test_colored_scatter_is_bitwise_reproducible(engine, oracle):
    """Like the reference's coloured path (disjoint subsets, colours in sequence) the COLORED strategy has a fixed
    summation order: repeated runs are bit-identical.  (ATOMIC and GATHER sum in hardware order and are only
    reproducible to rounding -- test_repeated_runs_agree.)"""
    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC")
    runs = [fa.CsrAssembler(fa.SCATTER_COLORED).assemble(asm).values for _ in range(3)]
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])


# ------------------------------------------------------------------------------------ mass matrix (row N1)
def _mass_pair(engine, oracle, kind, sdim, density=2.5):
    mesh = _mesh(kind)
    w, p = _rule(kind) if kind != "QUAD4" else quadrature.tensor.quadrilateral_gauss(3)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.Density(density))
    asm = fa.ElementMassAssembler.with_solution_dim(sdim, engine).with_space(mesh).with_quadrature_table(qt)
    d = mesh.vertices.shape[1]
    ref = oracle.ElementAssembler(KIND[kind], oracle.MASS_SCALAR if sdim == 1 else oracle.MASS_VECTOR, mesh.vertices,
                                  mesh.connectivity, w, p, params=(density, 0.0))
    assert sdim in (1, d)
    return asm, ref


def test_mass_matrix_reference_element_kat(engine):
    # tests/unit_tests/assembly/local.rs:38-69
    mesh = fa.Mesh(np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=float), np.array([[0, 1, 2, 3]]), fa.QUAD4)
    w, p = quadrature.tensor.quadrilateral_gauss(3)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.Density(3.0))
    asm = fa.ElementMassAssembler.with_solution_dim(2, engine).with_space(mesh).with_quadrature_table(qt)
    expected = np.kron(3.0 / 9.0 * np.array([[4, 2, 1, 2], [2, 4, 2, 1], [1, 2, 4, 2], [2, 1, 2, 4]], dtype=float), np.eye(2))
    assert np.allclose(asm.assemble_element_matrix(0), expected, rtol=0, atol=2e-15)


@pytest.mark.parametrize("kind,sdim", [("HEX8", 1), ("HEX8", 3), ("TET4", 3), ("QUAD4", 2), ("HEX27", 1), ("TRI3", 1)])
@pytest.mark.parametrize("scatter", [fa.SCATTER_ATOMIC, fa.SCATTER_COLORED, fa.SCATTER_GATHER])
def test_mass_matrix_matches_oracle(engine, oracle, kind, sdim, scatter):
    asm, ref = _mass_pair(engine, oracle, kind, sdim)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert st == 0
    k = fa.CsrAssembler(scatter).assemble(asm)
    assert np.array_equal(k.row_offsets, oro) and np.array_equal(k.col_indices, oci)
    assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max()
    with pytest.raises(fa.FenrisError):
        fa.VectorAssembler().assemble_vector(asm)


def test_host_path_overwrite_with_row_range_keeps_other_rows(engine, oracle):
    """fh_assemble_matrix (host arrays) with FH_ASSEMBLE_OVERWRITE and a row range: the rows in range are overwritten, every
    other entry of the caller's array keeps its value (include/fenris_hip.h, fh_set_row_range) -- also for an empty range."""
    mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(5)
    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC", mesh=mesh)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)       # builds the pattern
    n = mesh.num_nodes()
    for lo_n, hi_n in ((n // 3, 2 * n // 3), (7, 7)):
        engine.set_row_range(lo_n, hi_n)
        buf = np.full(len(vals), 123.25)
        engine.assemble_matrix(buf, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        lo, hi = int(ro[3 * lo_n]), int(ro[3 * hi_n])
        assert np.all(buf[:lo] == 123.25) and np.all(buf[hi:] == 123.25)
        if hi > lo:
            assert np.abs(buf[lo:hi] - vals[lo:hi]).max() <= TOL * np.abs(vals).max()
    engine.set_row_range(0, n)
    assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()


@pytest.mark.parametrize("kind,op", [("TET4", "LINEAR_ELASTIC"), ("TET4", "LAPLACE"), ("QUAD4", "LINEAR_ELASTIC"), ("TRI3", "LAPLACE"),
                                     ("HEX8", "NEO_HOOKEAN"), ("HEX27", "LINEAR_ELASTIC")])
@pytest.mark.parametrize("masked", [False, True])
def test_row_range_overwrite_on_the_device_other_kinds(engine, oracle, kind, op, masked):
    """fh_set_row_range + FH_ASSEMBLE_OVERWRITE into garbage, device arrays, for the kernels the Hex8 tests do not reach (the Tet4
    row-owner kernel, the planar pipelined forms, the two-pass path), with and without an element mask: rows in range = the atomic
    assembly's rows, everything else untouched"""
    import torch

    asm, ref = _pair(engine, oracle, kind, op)
    eng = asm.engine
    nnz = eng.build_pattern()
    ro, _ = eng.pattern(want_cols=False)
    s = asm.solution_dim()
    n = asm.num_nodes()
    if masked:
        eng.set_active_elements(np.arange(asm.num_elements()) % 4 != 2)
    try:
        want = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(want, fa.SCATTER_ATOMIC)
        w = want.cpu().numpy()
        for lo_n, hi_n in ((n // 3, 2 * n // 3), (0, 2), (n - 3, n), (5, 5)):
            eng.set_row_range(lo_n, hi_n)
            got = torch.full((nnz,), 99.5, dtype=torch.float64, device="cuda")
            eng.assemble_matrix(got, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
            g = got.cpu().numpy()
            lo, hi = int(ro[s * lo_n]), int(ro[s * hi_n])
            assert np.all(g[:lo] == 99.5) and np.all(g[hi:] == 99.5), (eng.last_kernel_name(), lo_n, hi_n)
            if hi > lo:
                assert np.abs(g[lo:hi] - w[lo:hi]).max() <= TOL * np.abs(w).max(), (eng.last_kernel_name(), lo_n, hi_n)
    finally:
        eng.set_row_range(0, n)
        eng.set_active_elements(None)


def test_quadrature_data_with_record_stride(engine, oracle):
    """fh_set_quadrature_uniform_data: the per-point Parameters where the host keeps them -- LameParameters inside a larger
    record (stride 32), Density records (stride 8) -- give what the packed pair layout gives"""
    import ctypes as C

    from fenris_amd import _ffi

    mesh = _mesh("HEX8")
    w, p = _rule("HEX8")
    nq = len(w)
    lam = np.array([[LAME.mu * (1 + 0.03 * q), LAME.lambda_ * (1 - 0.02 * q)] for q in range(nq)])
    asm, ref = _pair(engine, oracle, "HEX8", "LINEAR_ELASTIC", mesh=mesh)
    ref = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, mesh.vertices, mesh.connectivity, w, p, params=lam)
    _, _, ro, ci, vals = oracle.assemble(ref)
    recs = np.zeros((nq, 4))           # {mu, lambda, something else, padding}
    recs[:, :2] = lam
    recs[:, 2] = 123.0
    wv, pv = _ffi.as_f64(w), _ffi.as_f64(p)
    lib = _ffi.lib()
    engine._check(lib.fh_set_quadrature_uniform_data(engine._h, _ffi.fp(wv), _ffi.fp(pv), nq, recs.ctypes.data_as(C.c_void_p), 32, 1))
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()
    assert lib.fh_set_quadrature_uniform_data(engine._h, _ffi.fp(wv), _ffi.fp(pv), nq, recs.ctypes.data_as(C.c_void_p), 12, 1) == _ffi.FH_BAD_ARGUMENT
    assert lib.fh_set_quadrature_uniform_data(engine._h, _ffi.fp(wv), _ffi.fp(pv), nq, recs.ctypes.data_as(C.c_void_p), 8, 1) == _ffi.FH_BAD_ARGUMENT
