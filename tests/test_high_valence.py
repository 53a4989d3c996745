"""Meshes with nodes of unusual valence: fans of K elements around an edge / a vertex.  The fast owner-computes kernels carry per-node
budgets (terms per block and lanes per position in the Tet4 row-owner kernel, entries and slots per block in the pipelined kernel, unique
vertices per position): these meshes run into every one of them, so either the fast kernel must handle the case or the fallback chain
(smaller blocks -> standard tables -> generic gather kernel) must -- the result has to be the oracle's either way.  The reference has no
such limits (global.rs:133-182 walks any connectivity)."""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


def tet_fan(k, layers=1):
    """k tetrahedra (A, B, R_i, R_i+1) around the edge A-B, stacked `layers` times along the axis: the axis nodes have up to 2 k elements"""
    ang = np.linspace(0.0, 2.0 * np.pi, k, endpoint=False)
    ring = lambda z: np.stack([np.cos(ang) * (1.0 + 0.1 * np.sin(3 * ang)), np.sin(ang), np.full(k, z)], axis=1)
    verts = [np.array([[0.0, 0.0, float(z)]]) for z in range(layers + 1)]      # axis nodes 0 .. layers
    rings = [ring(z + 0.5) for z in range(layers)]
    v = np.concatenate(verts + rings)
    conn = []
    for z in range(layers):
        r0 = layers + 1 + z * k
        for i in range(k):
            conn.append([z, z + 1, r0 + i, r0 + (i + 1) % k])
    return fa.Mesh(v, np.asarray(conn, dtype=np.uint64), fa.TET4)


def quad_fan(k):
    """k quadrilaterals (C, P_2i, P_2i+1, P_2i+2) around the vertex C"""
    ang = np.linspace(0.0, 2.0 * np.pi, 2 * k, endpoint=False)
    rad = np.where(np.arange(2 * k) % 2 == 0, 1.0, 1.3)
    v = np.concatenate([np.zeros((1, 2)), np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1)])
    conn = [[0, 1 + 2 * i, 1 + (2 * i + 1) % (2 * k), 1 + (2 * i + 2) % (2 * k)] for i in range(k)]
    return fa.Mesh(v, np.asarray(conn, dtype=np.uint64), fa.QUAD4)


def hex_fan(k, layers=2):
    """the quadrilateral fan extruded: the axis nodes have up to 2 k hexahedra"""
    q = quad_fan(k)
    n2 = q.num_nodes()
    v = np.concatenate([np.concatenate([q.vertices, np.full((n2, 1), float(z))], axis=1) for z in range(layers + 1)])
    c2 = np.asarray(q.connectivity).astype(np.int64)
    conn = [list(c + z * n2) + list(c + (z + 1) * n2) for z in range(layers) for c in c2]
    return fa.Mesh(v, np.asarray(conn, dtype=np.uint64), fa.HEX8)


def _rule(kind):
    if kind == fa.QUAD4:
        return quadrature.tensor.quadrilateral_gauss(2)
    if kind == fa.HEX8:
        return quadrature.tensor.hexahedron_gauss(2)
    return quadrature.total_order.tetrahedron(2)


CASES = [("tet", 6, 1), ("tet", 30, 1), ("tet", 30, 3), ("tet", 44, 2), ("tet", 60, 1), ("tet", 130, 2),
         ("quad", 3, 0), ("quad", 5, 0), ("quad", 12, 0), ("quad", 40, 0), ("quad", 140, 0),
         ("hex", 3, 2), ("hex", 5, 2), ("hex", 12, 2), ("hex", 20, 3), ("hex", 70, 2)]


@pytest.mark.parametrize("shape,k,layers", CASES)
@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
def test_fans_match_oracle(engine, oracle, shape, k, layers, op):
    mesh = {"tet": lambda: tet_fan(k, layers), "quad": lambda: quad_fan(k), "hex": lambda: hex_fan(k, layers)}[shape]()
    okind = {"tet": oracle.TET4, "quad": oracle.QUAD4, "hex": oracle.HEX8}[shape]
    w, p = _rule(mesh.elem_kind)
    d = mesh.vertices.shape[1]
    s = 1 if op == "LAPLACE" else d
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    lame = fa.LameParameters(2.0e5, 3.0e5)
    if op != "LAPLACE":
        qt = qt.with_uniform_data(lame)
    operator = fa.LaplaceOperator() if op == "LAPLACE" else fa.MaterialEllipticOperator(fa.LinearElasticMaterial())
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(operator)
           .with_quadrature_table(qt).with_u(np.zeros(s * mesh.num_nodes())).build())
    ref = oracle.ElementAssembler(okind, getattr(oracle, op), mesh.vertices, mesh.connectivity, w, p,
                                  params=(lame.as_pair() if op != "LAPLACE" else None), u=np.zeros(s * mesh.num_nodes()))
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert st == 0
    for scatter in (fa.SCATTER_GATHER, fa.SCATTER_ATOMIC):
        kmat = fa.CsrAssembler(scatter).assemble(asm)
        assert np.array_equal(kmat.row_offsets, oro) and np.array_equal(kmat.col_indices, oci)
        err = np.abs(kmat.values - ovals).max() / np.abs(ovals).max()
        assert err <= TOL, (engine.last_kernel_name(), err)
    # residual and energy take the element pass whatever the valence
    f = fa.VectorAssembler().assemble_vector(asm)
    st, _, of = oracle.assemble_vector(ref)
    assert st == 0 and np.abs(f - of).max() <= TOL * max(np.abs(of).max(), 1.0)


@pytest.mark.parametrize("shape,k,layers", [("tet", 130, 2), ("quad", 140, 0), ("hex", 70, 2)])
def test_parallel_coloring_beyond_one_window(engine, oracle, shape, k, layers):
    """a node with more elements than the 128 colours one pass of fh_color_parallel marks: further windows, wider sort keys"""
    mesh = {"tet": lambda: tet_fan(k, layers), "quad": lambda: quad_fan(k), "hex": lambda: hex_fan(k, layers)}[shape]()
    w, p = _rule(mesh.elem_kind)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(fa.LaplaceOperator())
           .with_quadrature_table(qt).with_u(np.zeros(mesh.num_nodes())).build())
    colors = engine.color_parallel()
    conn = np.asarray(mesh.connectivity).astype(np.int64)
    assert len(colors) >= (2 * k if layers >= 2 else k)           # the elements at the busiest node are pairwise neighbours
    assert sorted(colors.labels.tolist()) == list(range(mesh.num_elements()))
    for c in range(len(colors)):
        lab = colors.color(c).astype(np.int64)
        nodes = conn[lab].ravel()
        assert np.all(np.diff(lab) > 0) and len(np.unique(nodes)) == len(nodes)
    a = fa.CsrAssembler(fa.SCATTER_ATOMIC).assemble(asm)
    c = fa.CsrParAssembler().assemble(colors, asm)
    assert np.abs(c.values - a.values).max() <= TOL * np.abs(a.values).max()


@pytest.mark.parametrize("n,ne,k,hubs", [(5000, 3000, 4, 1), (20000, 20000, 8, 3), (300, 5000, 3, 1)])
@pytest.mark.parametrize("s", [1, 3])
def test_pattern_with_hub_nodes(engine, n, ne, k, hubs, s):
    """assemble_pattern (global.rs:65-120 walks ANY connectivity) with nodes that sit in thousands of elements: more candidate neighbours than
    the per-node sort in LDS takes -- the bitmap path of pattern_kernels.hpp -- bit-exact against the pattern of A^T A"""
    import scipy.sparse as sp

    rng = np.random.default_rng(0)
    elems = [[int(rng.integers(0, hubs))] + [int(x) for x in rng.integers(hubs, n, k - 1)] for _ in range(ne)]
    mock = fa.MockElementAssembler(s, n, elems, engine)
    ro, ci = fa.CsrAssembler().assemble_pattern(mock)
    inc = sp.csr_matrix((np.ones(ne * k), (np.repeat(np.arange(ne), k), np.asarray(elems).ravel())), shape=(ne, n))
    pat = (inc.T @ inc).tocsr()
    pat.data[:] = 1.0
    full = sp.kron(pat, np.ones((s, s))).tocsr()
    full.sort_indices()
    assert np.array_equal(np.asarray(ro, dtype=np.int64), full.indptr) and np.array_equal(np.asarray(ci, dtype=np.int64), full.indices)


def test_two_thousand_tetrahedra_around_an_edge(engine, oracle):
    """past every budget at once: 8000 candidate neighbours at the axis nodes (bitmap pattern), rows of 2002 blocks, 2000 colours"""
    mesh = tet_fan(2000, 1)
    w, p = quadrature.total_order.tetrahedron(1)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(fa.LaplaceOperator())
           .with_quadrature_table(qt).with_u(np.zeros(mesh.num_nodes())).build())
    ref = oracle.ElementAssembler(oracle.TET4, oracle.LAPLACE, mesh.vertices, mesh.connectivity, w, p, u=np.zeros(mesh.num_nodes()))
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert st == 0
    for scatter in (fa.SCATTER_ATOMIC, fa.SCATTER_GATHER):
        k = fa.CsrAssembler(scatter).assemble(asm)
        assert np.array_equal(k.row_offsets, oro) and np.array_equal(k.col_indices, oci)
        assert np.abs(k.values - ovals).max() <= 1e-11 * np.abs(ovals).max()      # needle elements: cond(J) eps
    kc = fa.CsrParAssembler().assemble(fa.color_nodes(asm), asm)
    assert np.abs(kc.values - ovals).max() <= 1e-11 * np.abs(ovals).max()


def tet_bouquet(k):
    """k tetrahedra that share ONE vertex and nothing else (plus a regular neighbourhood: a small BCC box appended, so that the mesh also has
    ordinary nodes): the centre has 4 k candidate neighbours and 3 k + 1 distinct ones"""
    rng = np.random.default_rng(k)
    dirs = rng.standard_normal((k, 3))
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    verts = [np.zeros((1, 3))]
    conn = []
    for i in range(k):
        a = dirs[i]
        b = np.cross(a, [0.3, 0.5, 0.8])
        b /= np.linalg.norm(b)
        cc = np.cross(a, b)
        base = 1 + 3 * i
        verts.append(np.stack([a + 0.05 * b, a - 0.03 * b + 0.05 * cc, a - 0.03 * b - 0.05 * cc]))
        conn.append([0, base, base + 1, base + 2])
    box = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
    off = 1 + 3 * k
    v = np.concatenate(verts + [box.vertices + 3.0])
    c = np.concatenate([np.asarray(conn, dtype=np.int64), np.asarray(box.connectivity).astype(np.int64) + off])
    return fa.Mesh(v, c.astype(np.uint64), fa.TET4)


@pytest.mark.parametrize("k", [16, 21, 22, 30, 32, 33])
def test_pattern_one_pass_with_two_candidates_per_lane(engine, oracle, k):
    """the one-pass neighbour kernel with 65 ... 128 candidates per node (two per lane, wave_sort128) and its fall-back: k = 21 fills the scratch
    row exactly (64 distinct neighbours), 22 ... 32 overflow it (two passes instead), 33 has more than 128 candidates (two passes from the
    start); indices bit-exact and values against the oracle"""
    mesh = tet_bouquet(k)
    w, p = quadrature.total_order.tetrahedron(1)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(fa.LaplaceOperator())
           .with_quadrature_table(qt).with_u(np.zeros(mesh.num_nodes())).build())
    ref = oracle.ElementAssembler(oracle.TET4, oracle.LAPLACE, mesh.vertices, mesh.connectivity, w, p)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert st == 0
    assert oro[1] - oro[0] == 3 * k + 1
    kmat = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert np.array_equal(kmat.row_offsets, oro) and np.array_equal(kmat.col_indices, oci)
    assert np.abs(kmat.values - ovals).max() <= TOL * np.abs(ovals).max()
