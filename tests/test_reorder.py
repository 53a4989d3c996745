"""Reverse Cuthill-McKee reordering (src/mesh/reorder.rs): the reference's known answers, product == oracle, and the
properties the assembly path relies on.  Host code only (no device calls)."""
import numpy as np
import pytest
import scipy.sparse as sp

import fenris_amd as fa
from fenris_amd import quadrature, reorder
from conftest import load_golden_mesh


def _pattern(dense):
    m = sp.csr_matrix(np.array(dense))
    m.sort_indices()
    return m.indptr.astype(np.uint64), m.indices.astype(np.uint64)


def test_cuthill_mckee_basic_examples(oracle):
    """tests/unit_tests/reorder.rs:5-35"""
    ro, ci = _pattern([[1, 0, 1, 1], [0, 1, 0, 1], [1, 0, 1, 0], [1, 1, 0, 1]])
    for perm in (reorder.cuthill_mckee(ro, ci).perm(), oracle.cuthill_mckee(ro, ci)):
        assert list(perm) == [1, 3, 0, 2]
    rcm = reorder.reverse_cuthill_mckee(ro, ci)
    assert list(rcm.perm()) == [2, 0, 3, 1]
    ro, ci = _pattern(np.eye(4, dtype=int))
    for perm in (reorder.cuthill_mckee(ro, ci).perm(), oracle.cuthill_mckee(ro, ci)):
        assert list(perm) == [0, 1, 2, 3]


def test_permutation_type():
    p = reorder.Permutation.from_vec([2, 0, 1])
    assert p.source_index(0) == 2 and list(p.inverse().perm()) == [1, 2, 0]
    assert list(p.apply_to_slice(np.array([10, 20, 30]))) == [30, 10, 20]
    with pytest.raises(reorder.InvalidPermutation):
        reorder.Permutation.from_vec([0, 0, 1])
    with pytest.raises(ValueError):
        p.apply_to_slice(np.zeros(4))


def _shuffled(mesh, seed):
    rng = np.random.default_rng(seed)
    vp = rng.permutation(mesh.num_nodes())
    inv = np.empty_like(vp)
    inv[vp] = np.arange(len(vp))
    conn = inv[mesh.connectivity.astype(np.int64)][rng.permutation(mesh.num_elements())]
    return fa.Mesh(mesh.vertices[vp], conn.astype(np.uint64), mesh.elem_kind)


def _bandwidth(mesh):
    c = mesh.connectivity.astype(np.int64)
    return int((c.max(axis=1) - c.min(axis=1)).max())


@pytest.mark.parametrize("name", ["tet_bcc", "hex", "sphere_tet4_593", "square_quad4_79"])
def test_reorder_mesh_matches_oracle_and_restores_locality(oracle, name):
    if name == "tet_bcc":
        base = fa.procedural.create_unit_box_uniform_tet_mesh_3d(4)
    elif name == "hex":
        base = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 2, 1, 1, 4)
    else:
        v, c = load_golden_mesh(name)
        base = fa.Mesh(v, c, fa.TET4 if "tet4" in name else fa.QUAD4)
    mesh = _shuffled(base, 12345)
    mp = reorder.reorder_mesh_par(mesh)
    ovp, ocp = oracle.reorder_mesh(mesh.num_nodes(), mesh.connectivity)
    assert np.array_equal(mp.vertex_permutation().perm(), ovp)
    assert np.array_equal(mp.connectivity_permutation().perm(), ocp)
    # valid permutations
    assert sorted(ovp.tolist()) == list(range(mesh.num_nodes())) and sorted(ocp.tolist()) == list(range(mesh.num_elements()))
    new = mp.apply(mesh)
    # same geometry: every new element has the vertices of the old element it came from, in the same local order
    old_elems = mesh.vertices[mesh.connectivity.astype(np.int64)][ocp.astype(np.int64)]
    assert np.array_equal(new.vertices[new.connectivity.astype(np.int64)], old_elems)
    # locality: the shuffled numbering has an element index span of ~N, RCM brings it down to a front width
    assert _bandwidth(new) < 0.5 * _bandwidth(mesh)
    # elements follow their smallest vertex index
    mins = new.connectivity.astype(np.int64).min(axis=1)
    assert np.all(np.diff(mins) >= 0)


def test_reordered_mesh_assembles_the_permuted_matrix(oracle):
    """P K P^T: assembling the reordered mesh gives the original matrix with rows / columns permuted"""
    base = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
    mesh = _shuffled(base, 7)
    w, p = quadrature.total_order.tetrahedron(1)
    mp = reorder.reorder_mesh_par(mesh)
    new = mp.apply(mesh)
    a = oracle.ElementAssembler(oracle.TET4, oracle.LAPLACE, mesh.vertices, mesh.connectivity, w, p)
    b = oracle.ElementAssembler(oracle.TET4, oracle.LAPLACE, new.vertices, new.connectivity, w, p)
    _, _, ro, ci, va = oracle.assemble(a)
    _, _, rob, cib, vb = oracle.assemble(b)
    n = mesh.num_nodes()
    ka = sp.csr_matrix((va, ci.astype(np.int64), ro.astype(np.int64)), shape=(n, n)).toarray()
    kb = sp.csr_matrix((vb, cib.astype(np.int64), rob.astype(np.int64)), shape=(n, n)).toarray()
    src = mp.vertex_permutation().perm().astype(np.int64)  # new index t <- old index src[t]
    assert np.abs(kb - ka[np.ix_(src, src)]).max() <= 1e-13 * np.abs(ka).max()
