"""Pins the CPU oracle (oracle/fenris_oracle.c) against every known-answer test, snapshot and golden
value the reference's own test-suite holds for the assembly path.

Citations are file:line relative to the reference checkout.
"""
import itertools
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden_mesh


# ------------------------------------------------------------------ integer KATs: sparsity pattern
# tests/unit_tests/assembly/global.rs:70-142 (serial) and :144-216 (parallel, same numbers)
MOCK_CONN = [[0, 1, 2], [2, 3], [], [3, 4, 4, 4, 4, 4, 4]]


def _ragged(conn):
    offs = np.cumsum([0] + [len(c) for c in conn]).astype(np.uint64)
    nodes = np.array([x for c in conn for x in c], dtype=np.uint64)
    return offs, nodes


def test_pattern_kat_empty(oracle):
    ro, ci = oracle.assemble_pattern(1, 0, *_ragged([[]]))
    assert ro.tolist() == [0] and ci.tolist() == []
    ro, ci = oracle.assemble_pattern(2, 5, *_ragged([[]]))
    assert ro.tolist() == [0] * 11 and ci.tolist() == []


def test_pattern_kat_sdim1(oracle):
    ro, ci = oracle.assemble_pattern(1, 6, *_ragged(MOCK_CONN))
    assert ro.tolist() == [0, 3, 6, 10, 13, 15, 15]  # global.rs:109-116
    assert ci.tolist() == [0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 2, 3, 4, 3, 4]


def test_pattern_kat_sdim2(oracle):
    ro, ci = oracle.assemble_pattern(2, 6, *_ragged(MOCK_CONN))
    assert ro.tolist() == [0, 6, 12, 18, 24, 32, 40, 46, 52, 56, 60, 60, 60]  # global.rs:128-137
    assert ci.tolist() == [
        0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 6, 7, 0, 1,
        2, 3, 4, 5, 6, 7, 4, 5, 6, 7, 8, 9, 4, 5, 6, 7, 8, 9, 6, 7, 8, 9, 6, 7, 8, 9]


# ------------------------------------------------------------------ mesh generator snapshots
@pytest.mark.parametrize("res", [1, 2])
def test_bcc_tet_mesh_snapshot(oracle, res):
    # tests/unit_tests/mesh/procedural.rs:18-30 + snapshots mesh_{1,2}.snap
    gv, gc = load_golden_mesh(f"tet_mesh_res{res}")
    v, c = oracle.tet_mesh(1.0, 1, 1, 1, res)
    assert np.array_equal(c, gc)
    assert np.array_equal(v, gv)
    assert len(c) == 12 * res ** 3


def test_hex_mesh_layout(oracle):
    # src/mesh/procedural.rs:241-271
    v, c = oracle.unit_box_hex_mesh(2)
    assert v.shape == (27, 3) and c.shape == (8, 8)
    assert c[0].tolist() == [0, 1, 4, 3, 9, 10, 13, 12]
    assert v[1].tolist() == [0.5, 0.0, 0.0] and v[3].tolist() == [0.0, 0.5, 0.0] and v[9].tolist() == [0.0, 0.0, 0.5]


def test_quad_mesh_layout(oracle):
    # src/mesh/procedural.rs:68-89: vertices row by row from top-left (0, 1)
    v, c = oracle.unit_square_quad_mesh(2)
    assert v.shape == (9, 2) and c.shape == (4, 4)
    assert v[0].tolist() == [0.0, 1.0] and v[2].tolist() == [1.0, 1.0] and v[8].tolist() == [1.0, 0.0]
    assert c[0].tolist() == [3, 4, 1, 0]


def test_hex27_conversion(oracle):
    # src/mesh_convert.rs:85-166, 283-327: first-occurrence numbering, local nodes 0..26 of element 0 first
    v8, c8 = oracle.unit_box_hex_mesh(2)
    v27, c27 = oracle.hex8_to_hex27(v8, c8)
    assert c27[0].tolist() == list(range(27))
    assert len(v27) == 125  # 5^3 unique nodes
    assert len(np.unique(np.round(v27, 12), axis=0)) == 125
    # node positions agree with the Hex27 reference-node table mapped through the trilinear map
    ref = np.array([oracle.element_basis(oracle.HEX8, s) @ v8[c8[0].astype(int)] for s in _hex27_nodes()])
    assert np.allclose(v27[c27[0].astype(int)], ref, atol=1e-15)


def _hex27_nodes():
    # src/element/hexahedron.rs:179-210
    return np.array([
        [-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1],
        [0, -1, -1], [-1, 0, -1], [-1, -1, 0], [1, 0, -1], [1, -1, 0], [0, 1, -1], [1, 1, 0], [-1, 1, 0],
        [0, -1, 1], [-1, 0, 1], [1, 0, 1], [0, 1, 1],
        [0, 0, -1], [0, -1, 0], [-1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 0]], dtype=float)


# ------------------------------------------------------------------ material KATs
def test_lame_from_young_poisson(oracle):
    # fenris-solid/tests/unit_tests/materials.rs:74-85
    mu, lam = oracle.lame_from_young_poisson(1e3, 0.3)
    assert mu == pytest.approx(384.6153846153846, rel=4e-16)
    assert lam == pytest.approx(576.9230769230769, rel=4e-16)


F2 = np.array([[2.0, 1.0], [3.0, 4.0]])                       # fenris-solid/tests/unit_tests/mod.rs:18-22
F3 = np.array([[2.0, 1.0, 3.0], [4.0, 6.0, 5.0], [2.0, 8.0, 9.0]])  # mod.rs:24-29
MU, LAM = 384.0, 577.0                                          # mod.rs:11-16


@pytest.mark.parametrize("op,F,expected", [
    ("LINEAR_ELASTIC", F2, 11528.0),            # materials.rs:246-252
    ("LINEAR_ELASTIC", F3, 133154.0),           # materials.rs:255-262
    ("STVK", F2, 132578.0),                     # materials.rs:299-306
    ("STVK", F3, 9136789.125),                  # materials.rs:308-315
    ("NEO_HOOKEAN", F2, 5505.274620288603),     # materials.rs:335-342
    ("NEO_HOOKEAN", F3, 48833.26962613859),     # materials.rs:344-351
])
def test_material_energy_kat(oracle, op, F, expected):
    psi = oracle.material_energy_density(getattr(oracle, op), F, MU, LAM)
    assert psi == pytest.approx(expected, rel=1e-14)


@pytest.mark.parametrize("op", ["LINEAR_ELASTIC", "NEO_HOOKEAN", "STVK"])
@pytest.mark.parametrize("F", [F2, F3])
def test_stress_is_derivative_of_energy(oracle, op, F):
    # fenris-solid/tests/unit_tests/materials.rs:87-120 (central FD, h = 1e-5... tol relative to max)
    opk = getattr(oracle, op)
    P = oracle.material_stress_tensor(opk, F, MU, LAM)
    h = 1e-6
    P_fd = np.zeros_like(F)
    for i, j in itertools.product(range(F.shape[0]), repeat=2):
        Fp, Fm = F.copy(), F.copy()
        Fp[i, j] += h
        Fm[i, j] -= h
        P_fd[i, j] = (oracle.material_energy_density(opk, Fp, MU, LAM)
                      - oracle.material_energy_density(opk, Fm, MU, LAM)) / (2 * h)
    assert np.allclose(P, P_fd, rtol=0, atol=1e-6 * np.abs(P).max())


@pytest.mark.parametrize("op", ["LINEAR_ELASTIC", "NEO_HOOKEAN", "STVK"])
@pytest.mark.parametrize("F", [F2, F3])
def test_contraction_consistent_with_stress(oracle, op, F):
    # materials.rs:122-168: C_ij = a_k dP_ik/dF_jl b_l by central differences
    opk = getattr(oracle, op)
    d = F.shape[0]
    rng = np.random.default_rng(7)
    a, b = rng.standard_normal(d), rng.standard_normal(d)
    Cm = oracle.material_stress_contraction(opk, F, a, b, MU, LAM)
    h = 1e-5
    C_fd = np.zeros((d, d))
    for j, l in itertools.product(range(d), repeat=2):
        Fp, Fm = F.copy(), F.copy()
        Fp[j, l] += h
        Fm[j, l] -= h
        dP = (oracle.material_stress_tensor(opk, Fp, MU, LAM) - oracle.material_stress_tensor(opk, Fm, MU, LAM)) / (2 * h)
        for i, k in itertools.product(range(d), repeat=2):
            C_fd[i, j] += a[k] * dP[i, k] * b[l]
    assert np.allclose(Cm, C_fd, rtol=0, atol=1e-7 * np.abs(C_fd).max())


def test_neo_hookean_nonpositive_J_is_nan(oracle):
    # materials.rs:298-300
    F = np.diag([1.0, -1.0, 1.0])
    Cm = oracle.material_stress_contraction(oracle.NEO_HOOKEAN, F, np.ones(3), np.ones(3), MU, LAM)
    assert np.isnan(Cm).all()
    assert np.isnan(oracle.material_stress_tensor(oracle.NEO_HOOKEAN, F, MU, LAM)).all()


# ------------------------------------------------------------------ quadrature
def test_gauss_point_order_and_values(oracle):
    # fenris-quadrature/src/univariate.rs:79-112: positive roots first (descending), then mirrored
    w, x = oracle.gauss(2)
    assert x[0] == pytest.approx(1 / np.sqrt(3), rel=1e-15) and x[1] == -x[0]
    assert np.allclose(w, [1, 1], rtol=1e-15)
    w, x = oracle.gauss(3)
    assert x[0] == pytest.approx(np.sqrt(3 / 5), rel=1e-15) and abs(x[1]) < 1e-16 and x[2] == -x[0]
    assert np.allclose(w, [5 / 9, 8 / 9, 5 / 9], rtol=1e-14)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 10, 20, 50, 200])
def test_gauss_integrates_monomials(oracle, n):
    # fenris-quadrature/tests/unit_tests/univariate.rs:6-26: exact to 1e-14 up to degree 2n-1
    w, x = oracle.gauss(n)
    for deg in range(0, min(2 * n, 40)):
        exact = 0.0 if deg % 2 else 2.0 / (deg + 1)
        assert abs(np.sum(w * x ** deg) - exact) < 1e-14


def test_tensor_rules_order(oracle):
    # tensor.rs:21-27,44-52: x outermost, last coordinate innermost
    w1, x1 = oracle.gauss(2)
    w, p = oracle.hexahedron_gauss(2)
    assert np.allclose(p[0], [x1[0]] * 3) and np.allclose(p[1], [x1[0], x1[0], x1[1]])
    assert np.allclose(p[4], [x1[1], x1[0], x1[0]])
    assert w.sum() == pytest.approx(8.0, rel=1e-15)
    w, p = oracle.quadrilateral_gauss(3)
    assert np.allclose(p[1], [oracle.gauss(3)[1][0], oracle.gauss(3)[1][1]])
    assert w.sum() == pytest.approx(4.0, rel=1e-15)


def test_simplex_rules(oracle):
    # rules/polyquad/expanded/tet/1-1.txt, 2-4.txt, 3-8.txt ; tri/1-1.txt, 2-3.txt
    w, p = oracle.tetrahedron_rule(1)
    assert p.tolist() == [[-0.5, -0.5, -0.5]] and w[0] == pytest.approx(4 / 3, rel=1e-16)
    for s, npts in [(2, 4), (3, 8)]:
        w, p = oracle.tetrahedron_rule(s)
        assert len(w) == npts and w.sum() == pytest.approx(4 / 3, rel=1e-15)
        # integrates x*y exactly over the reference tet (vertices (-1,-1,-1),(1,-1,-1),(-1,1,-1),(-1,-1,1))
        assert np.sum(w * p[:, 0]) == pytest.approx(-2 / 3, rel=1e-14)
    w, p = oracle.triangle_rule(2)
    assert w.sum() == pytest.approx(2.0, rel=1e-15)


# ------------------------------------------------------------------ element properties
@pytest.mark.parametrize("kind", ["QUAD4", "HEX8", "TET4", "HEX27", "TRI3"])
def test_partition_of_unity_and_gradient_sum(oracle, kind):
    # tests/unit_tests/element.rs:132-231
    k = getattr(oracle, kind)
    d = oracle.element_dim(k)
    rng = np.random.default_rng(3)
    for _ in range(5):
        xi = rng.uniform(-1, 1, d) * 0.5 - 0.25
        assert oracle.element_basis(k, xi).sum() == pytest.approx(1.0, abs=1e-14)
        assert np.abs(oracle.element_gradients(k, xi).sum(axis=1)).max() < 1e-14


def test_hex_lagrange_property_and_gradients_fd(oracle):
    # tests/unit_tests/element/hexahedron.rs:18-67 (Lagrange), :149-190 (gradients vs FD)
    nodes = _hex27_nodes()
    for kind, n in [(oracle.HEX8, 8), (oracle.HEX27, 27)]:
        for i in range(n):
            phi = oracle.element_basis(kind, nodes[i])
            e = np.zeros(n)
            e[i] = 1
            assert np.allclose(phi, e, atol=1e-15)
        xi = np.array([0.3, -0.2, 0.55])
        g = oracle.element_gradients(kind, xi)
        h = 1e-6
        for c in range(3):
            dx = np.zeros(3)
            dx[c] = h
            fd = (oracle.element_basis(kind, xi + dx) - oracle.element_basis(kind, xi - dx)) / (2 * h)
            assert np.allclose(g[c], fd, atol=1e-9)


# ------------------------------------------------------------------ element matrices
def test_quad4_reference_laplace_element_matrix(oracle):
    # tests/unit_tests/assembly.rs:159-162 (analytic reference-element Laplace stiffness)
    verts = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=float)
    conn = np.array([[0, 1, 2, 3]], dtype=np.uint64)
    w, p = oracle.quadrilateral_gauss(2)
    asm = oracle.ElementAssembler(oracle.QUAD4, oracle.LAPLACE, verts, conn, w, p)
    st, ke = asm.element_matrix(0)
    expected = np.array([[2 / 3, -1 / 6, -1 / 3, -1 / 6], [-1 / 6, 2 / 3, -1 / 6, -1 / 3],
                         [-1 / 3, -1 / 6, 2 / 3, -1 / 6], [-1 / 6, -1 / 3, -1 / 6, 2 / 3]])
    assert st == 0
    assert np.allclose(ke, expected, rtol=0, atol=1e-15)


def _distorted_hex(rng):
    ref = _hex27_nodes()[:8]
    return ref * np.array([1.0, 1.3, 0.8]) + rng.uniform(-0.15, 0.15, (8, 3)) + np.array([2.0, 0.5, 1.0])


def _single_element_assembler(oracle, kind, op, verts, u, rule, params=(MU, LAM)):
    n = oracle.element_num_nodes(kind)
    conn = np.arange(n, dtype=np.uint64)[None, :]
    w, p = rule
    return oracle.ElementAssembler(kind, op, verts, conn, w, p, params=None if op == oracle.LAPLACE else params, u=u)


ELEMENT_CASES = [
    ("TET4", lambda o: o.tetrahedron_rule(2)),
    ("HEX8", lambda o: o.hexahedron_gauss(2)),
    ("HEX27", lambda o: o.hexahedron_gauss(3)),
    ("QUAD4", lambda o: o.quadrilateral_gauss(2)),
]


@pytest.mark.parametrize("kind,rule", ELEMENT_CASES)
@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC", "NEO_HOOKEAN", "STVK"])
def test_element_vector_and_matrix_are_derivatives(oracle, kind, rule, op):
    """f_e = dE/du and K_e = df_e/du by central differences -- the reference's
    tests/unit_tests/assembly/local/elliptic.rs:281-455 and
    fenris-solid/tests/unit_tests/material_elliptic_operator.rs:76-173 (tol 1e-6)."""
    k, opk = getattr(oracle, kind), getattr(oracle, op)
    rng = np.random.default_rng(11)
    d = oracle.element_dim(k)
    n = oracle.element_num_nodes(k)
    s = oracle.solution_dim(opk, d)
    if kind == "TET4":
        verts = np.array([[2.0, 0, 1], [3, 4, 1], [1, 1, 2], [3, 1, 4]])  # elliptic.rs test element corners
    elif kind == "HEX8":
        verts = _distorted_hex(rng)
    elif kind == "HEX27":
        v8 = _distorted_hex(rng)
        verts = np.array([oracle.element_basis(oracle.HEX8, x) @ v8 for x in _hex27_nodes()])
    else:
        verts = np.array([[0.0, 0.1], [1.2, -0.1], [1.0, 1.1], [-0.1, 0.9]])
    u0 = 0.05 * rng.standard_normal(s * n)
    r = rule(oracle)

    def energy(u):
        st, e = _single_element_assembler(oracle, k, opk, verts, u, r).element_scalar(0)
        assert st == 0
        return e

    def vec(u):
        st, f = _single_element_assembler(oracle, k, opk, verts, u, r).element_vector(0)
        assert st == 0
        return f

    st, ke = _single_element_assembler(oracle, k, opk, verts, u0, r).element_matrix(0)
    assert st == 0
    f0 = vec(u0)
    h = 1e-6
    f_fd = np.zeros_like(f0)
    k_fd = np.zeros_like(ke)
    for i in range(s * n):
        du = np.zeros(s * n)
        du[i] = h
        f_fd[i] = (energy(u0 + du) - energy(u0 - du)) / (2 * h)
        k_fd[:, i] = (vec(u0 + du) - vec(u0 - du)) / (2 * h)
    assert np.allclose(f0, f_fd, rtol=0, atol=1e-6 * max(1.0, np.abs(f0).max()))
    assert np.allclose(ke, k_fd, rtol=0, atol=1e-6 * np.abs(ke).max())
    assert np.array_equal(ke, ke.T)  # clone_upper_to_lower makes K_e exactly symmetric (util.rs:46-50)


def test_singular_jacobian_error(oracle):
    # elliptic.rs:401-404: det exactly zero -> error
    verts = np.zeros((8, 3))
    asm = _single_element_assembler(oracle, oracle.HEX8, oracle.LAPLACE, verts, None, oracle.hexahedron_gauss(2))
    st, _ = asm.element_matrix(0)
    assert st == oracle.SINGULAR_JACOBIAN


# ------------------------------------------------------------------ colouring + global assembly
def test_coloring_properties(oracle):
    # fenris-paradis/src/coloring.rs:81-108 (proptest restated with a seeded generator)
    rng = np.random.default_rng(5)
    for _ in range(50):
        conn = [rng.integers(0, 100, rng.integers(0, 10)).tolist() for _ in range(rng.integers(0, 10))]
        offs, nodes = _ragged(conn)
        co, labels = oracle.color_elements(offs, nodes)
        assert len(co) - 1 <= len(conn)
        assert sorted(labels.tolist()) == list(range(len(conn)))
        for c in range(len(co) - 1):
            seen = set()
            for e in labels[co[c]:co[c + 1]]:
                s = set(conn[e])
                assert not (s & seen)
                seen |= s
            # order preserved inside a colour
            assert list(labels[co[c]:co[c + 1]]) == sorted(labels[co[c]:co[c + 1]])


def test_structured_hex_mesh_has_8_colors(oracle):
    v, c = oracle.unit_box_hex_mesh(4)
    offs = np.arange(0, 8 * (len(c) + 1), 8, dtype=np.uint64)
    co, labels = oracle.color_elements(offs, c.reshape(-1))
    assert len(co) - 1 == 8


def _csr_dense(ro, ci, vals, n):
    A = np.zeros((n, n))
    for r in range(n):
        for k in range(int(ro[r]), int(ro[r + 1])):
            A[r, int(ci[k])] += vals[k]
    return A


@pytest.mark.parametrize("kind,op,mesh,rule", [
    ("HEX8", "LAPLACE", lambda o: o.unit_box_hex_mesh(3), lambda o: o.hexahedron_gauss(2)),
    ("HEX8", "LINEAR_ELASTIC", lambda o: o.unit_box_hex_mesh(3), lambda o: o.hexahedron_gauss(2)),
    ("TET4", "LINEAR_ELASTIC", lambda o: o.unit_box_tet_mesh(2), lambda o: o.tetrahedron_rule(1)),
    ("QUAD4", "LAPLACE", lambda o: o.unit_square_quad_mesh(5), lambda o: o.quadrilateral_gauss(2)),
])
def test_serial_equals_colored_assembly(oracle, kind, op, mesh, rule):
    # tests/convergence_tests/poisson_mms_common.rs:102-121 (assert_matrix_eq comp = float)
    v, c = mesh(oracle)
    w, p = rule(oracle)
    asm = oracle.ElementAssembler(getattr(oracle, kind), getattr(oracle, op), v, c, w, p,
                                  params=None if op == "LAPLACE" else oracle.lame_from_young_poisson(1e6, 0.2))
    st, failed, ro, ci, vals = oracle.assemble(asm)
    assert st == 0
    colors = oracle.color_nodes(asm)
    vals2 = np.zeros_like(vals)
    st, _ = oracle.par_assemble_into_csr(asm, colors, ro, ci, vals2, num_threads=4)
    assert st == 0
    assert np.allclose(vals, vals2, rtol=0, atol=8 * np.finfo(float).eps * np.abs(vals).max())
    # dense cross-check: K = sum_e P_e^T K_e P_e
    n = asm.s * asm.N
    A = np.zeros((n, n))
    for e in range(asm.E):
        _, ke = asm.element_matrix(e)
        dofs = (asm.s * c[e].astype(int)[:, None] + np.arange(asm.s)[None, :]).reshape(-1)
        A[np.ix_(dofs, dofs)] += ke
    assert np.allclose(_csr_dense(ro, ci, vals, n), A, rtol=0, atol=1e-13 * np.abs(A).max())
    # vector path: serial == coloured
    u = np.random.default_rng(0).standard_normal(n) * 1e-3
    asm_u = oracle.ElementAssembler(getattr(oracle, kind), getattr(oracle, op), v, c, w, p, params=asm.params, u=u)
    st, _, f1 = oracle.assemble_vector(asm_u)
    st2, _, f2 = oracle.par_assemble_vector(asm_u, colors, num_threads=4)
    assert st == 0 and st2 == 0
    assert np.allclose(f1, f2, rtol=0, atol=1e-13 * np.abs(f1).max())
    if op in ("LAPLACE", "LINEAR_ELASTIC"):  # linear operators: f(u) = K u
        assert np.allclose(f1, A @ u, rtol=0, atol=1e-12 * np.abs(f1).max())


def test_assemble_accumulates(oracle):
    # global.rs:133-182: assemble_into_csr adds to the existing values
    v, c = oracle.unit_box_hex_mesh(2)
    asm = oracle.ElementAssembler(oracle.HEX8, oracle.LAPLACE, v, c, *oracle.hexahedron_gauss(2))
    st, _, ro, ci, vals = oracle.assemble(asm)
    vals2 = vals.copy()
    oracle.assemble_into_csr(asm, ro, ci, vals2)
    assert np.allclose(vals2, 2 * vals, rtol=1e-15)


def test_dirichlet_bc_csr_kat(oracle):
    # tests/unit_tests/assembly/global.rs:42-69
    n = 8
    ro = np.arange(0, n * n + 1, n, dtype=np.uint64)
    ci = np.tile(np.arange(n, dtype=np.uint64), n)
    vals = np.full(n * n, 2.0)
    oracle.apply_homogeneous_dirichlet_bc_csr(ro, ci, vals, [0, 2], 2)
    expected = np.array([
        2, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 0, 0, 2, 2, 0, 0, 2, 2, 0, 0, 2, 2,
        0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 2, 2, 0, 0, 2, 2, 0, 0, 2, 2, 0, 0, 2, 2], dtype=float)
    assert vals.tolist() == expected.tolist()


# ------------------------------------------------------------------ end-to-end golden: Poisson MMS
def _mms_errors(oracle, kind, v, c, rule, err_rule):
    """tests/convergence_tests/poisson_mms_common.rs:68-205: K from the oracle's Laplace assembly; the
    source RHS (local/source.rs), BC, solve and L2/H1 error integrals are restated in numpy/scipy here."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    d = v.shape[1]
    w, p = rule
    asm = oracle.ElementAssembler(kind, oracle.LAPLACE, v, c, w, p)
    st, _, ro, ci, vals = oracle.assemble(asm)
    assert st == 0
    N = len(v)
    u_exact = lambda x: np.prod(np.sin(np.pi * x), axis=-1)
    f = lambda x: d * np.pi ** 2 * u_exact(x)
    gkind = oracle.HEX8 if kind == oracle.HEX27 else kind
    ng = oracle.element_num_nodes(gkind)

    def geometry(e, xi):
        X = v[c[e, :ng].astype(int)]
        G = oracle.element_gradients(gkind, xi)
        J = X.T @ G.T
        x = oracle.element_basis(gkind, xi) @ X
        return x, J

    b = np.zeros(N)
    for e in range(len(c)):
        for wq, xi in zip(w, p):
            x, J = geometry(e, xi)
            b[c[e].astype(int)] += wq * abs(np.linalg.det(J)) * f(x) * oracle.element_basis(kind, xi)
    bc = np.where(np.abs(v - 0.5).max(axis=1) > 0.4999)[0]
    oracle.apply_homogeneous_dirichlet_bc_csr(ro, ci, vals, bc, 1)
    b[bc] = 0.0
    A = sp.csr_matrix((vals, ci.astype(np.int64), ro.astype(np.int64)), shape=(N, N))
    u_h = spla.spsolve(A.tocsc(), b)
    we, pe = err_rule
    l2 = h1 = 0.0
    for e in range(len(c)):
        ue = u_h[c[e].astype(int)]
        for wq, xi in zip(we, pe):
            x, J = geometry(e, xi)
            dv = wq * abs(np.linalg.det(J))
            l2 += dv * (oracle.element_basis(kind, xi) @ ue - u_exact(x)) ** 2
            grad_h = np.linalg.solve(J.T, oracle.element_gradients(kind, xi) @ ue)
            gx = np.array([np.pi * np.cos(np.pi * x[k]) * np.prod(np.sin(np.pi * np.delete(x, k))) for k in range(d)])
            h1 += dv * np.sum((grad_h - gx) ** 2)
    return np.sqrt(l2), np.sqrt(h1)


@pytest.mark.parametrize("name,kind,mesh,rule,err_rule,nres", [
    ("poisson2d_mms_quad4_summary", "QUAD4", lambda o, r: o.unit_square_quad_mesh(r),
     lambda o: o.quadrilateral_gauss(2), lambda o: o.quadrilateral_gauss(6), 5),
    ("poisson3d_mms_hex8_summary", "HEX8", lambda o, r: o.unit_box_hex_mesh(r),
     lambda o: o.hexahedron_gauss(2), lambda o: o.hexahedron_gauss(6), 3),
    ("poisson3d_mms_hex27_summary", "HEX27", lambda o, r: o.hex8_to_hex27(*o.unit_box_hex_mesh(r)),
     lambda o: o.hexahedron_gauss(4), lambda o: o.hexahedron_gauss(6), 3),
])
def test_mms_errors_match_reference_values(oracle, name, kind, mesh, rule, err_rule, nres):
    # tests/convergence_tests/poisson_{2d,3d}_mms.rs + reference_values/*.json, tolerance 1 %
    # (poisson_mms_common.rs:40-65)
    ref = json.load(open(os.path.join(GOLDEN, "mms_reference_values.json")))["summaries"][name]
    resolutions = [1, 2, 4, 8, 16, 32][:nres]
    for i, res in enumerate(resolutions):
        v, c = mesh(oracle, res)
        l2, h1 = _mms_errors(oracle, getattr(oracle, kind), v, c, rule(oracle), err_rule(oracle))
        assert abs(l2 - ref["L2_errors"][i]) / ref["L2_errors"][i] < 0.01, (res, l2, ref["L2_errors"][i])
        assert abs(h1 - ref["H1_seminorm_errors"][i]) / ref["H1_seminorm_errors"][i] < 0.01


def test_mms_tet4_matches_reference_values(oracle):
    # poisson_3d_mms.rs:118-126: quadrature total_order::tetrahedron(0) -> smallest tabulated rule (1 point);
    # error quadrature tetrahedron(6) = table tet/6-24.txt (golden fixture tet_rule_6_24.json)
    ref = json.load(open(os.path.join(GOLDEN, "mms_reference_values.json")))["summaries"]["poisson3d_mms_tet4_summary"]
    t = json.load(open(os.path.join(GOLDEN, "tet_rule_6_24.json")))
    err_rule = (np.array(t["weights"]), np.array(t["points"]))
    for i, res in enumerate([1, 2, 4]):
        v, c = oracle.unit_box_tet_mesh(res)
        l2, h1 = _mms_errors(oracle, oracle.TET4, v, c, oracle.tetrahedron_rule(1), err_rule)
        assert abs(l2 - ref["L2_errors"][i]) / ref["L2_errors"][i] < 0.01, (res, l2)
        assert abs(h1 - ref["H1_seminorm_errors"][i]) / ref["H1_seminorm_errors"][i] < 0.01, (res, h1)


# ------------------------------------------------------------------ mass matrix (next row N1)
def test_quad4_reference_mass_matrix_kat(oracle):
    # tests/unit_tests/assembly/local.rs:38-69: M = rho/9 [[4,2,1,2],[2,4,2,1],[1,2,4,2],[2,1,2,4]] (x) I_2
    verts = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=float)
    conn = np.array([[0, 1, 2, 3]], dtype=np.uint64)
    w, p = oracle.quadrilateral_gauss(3)  # exact for the bi-quadratic integrand, like the strength-5 rule of the test
    asm = oracle.ElementAssembler(oracle.QUAD4, oracle.MASS_VECTOR, verts, conn, w, p, params=(3.0, 0.0))
    st, me = asm.element_matrix(0)
    assert st == 0
    expected = np.kron(3.0 / 9.0 * np.array([[4, 2, 1, 2], [2, 4, 2, 1], [1, 2, 4, 2], [2, 1, 2, 4]], dtype=float), np.eye(2))
    assert np.allclose(me, expected, rtol=0, atol=4e-16 * 4)


def test_mass_matrix_total_mass(oracle):
    # sum of all entries of the scalar mass matrix = rho * volume
    v, c = oracle.unit_box_hex_mesh(3)
    w, p = oracle.hexahedron_gauss(2)
    asm = oracle.ElementAssembler(oracle.HEX8, oracle.MASS_SCALAR, v, c, w, p, params=(2.5, 0.0))
    st, _, ro, ci, vals = oracle.assemble(asm)
    assert st == 0 and vals.sum() == pytest.approx(2.5, rel=1e-13)
