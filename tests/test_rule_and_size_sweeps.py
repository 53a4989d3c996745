"""Sweeps that widen the parity net sideways (each case is small; the oracle is the checker):
  * every tensor Gauss rule up to 5 points per direction and every tabulated tetrahedron rule, on distorted and on affine meshes
    (the compile-time-rule instantiations, the chunked staging of long rules, the one-point collapse of the Tet4 row-owner kernel);
  * meshes of one, two and three elements of every kind with every operator and scatter (positions < workgroups, empty ranges);
  * structured meshes whose node and element numbering is randomly permuted (nothing is consecutive)."""
import itertools

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

pytestmark = pytest.mark.gpu
TOL = 1e-12
LAME = fa.LameParameters(3.0e2, 5.0e2)
OPS = {"LAPLACE": lambda: fa.LaplaceOperator(), "LINEAR_ELASTIC": lambda: fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
       "NEO_HOOKEAN": lambda: fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()), "STVK": lambda: fa.MaterialEllipticOperator(fa.StVKMaterial())}


@pytest.fixture(scope="module")
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


def _distorted(m, h, rng):
    return fa.Mesh(m.vertices + rng.uniform(-0.1 * h, 0.1 * h, m.vertices.shape), m.connectivity, m.elem_kind)


def _both(engine, oracle, mesh, okind, opname, w, p, u=None):
    d = mesh.vertices.shape[1]
    s = 1 if opname == "LAPLACE" else d
    u = np.zeros(s * mesh.num_nodes()) if u is None else u
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if opname != "LAPLACE":
        qt = qt.with_uniform_data(LAME)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(OPS[opname]()).with_quadrature_table(qt)
           .with_u(u).build())
    ref = oracle.ElementAssembler(okind, getattr(oracle, opname), mesh.vertices, mesh.connectivity, w, p,
                                  params=(LAME.as_pair() if opname != "LAPLACE" else None), u=u)
    return asm, ref


def _rules():
    out = []
    for n in range(1, 6):
        out += [("HEX8", f"gauss{n}", False), ("HEX8", f"gauss{n}", True), ("QUAD4", f"gauss{n}", False)]
    out += [("TET4", f"order{o}", False) for o in range(0, 7)]
    return out


@pytest.mark.parametrize("kind,rule,affine", _rules())
@pytest.mark.parametrize("opname", ["LAPLACE", "LINEAR_ELASTIC"])
def test_every_rule(engine, oracle, kind, rule, affine, opname):
    rng = np.random.default_rng(11)
    n = int(rule[-1])
    if kind == "HEX8":
        mesh, okind, (w, p) = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 3), oracle.HEX8, quadrature.tensor.hexahedron_gauss(n)
        mesh = mesh if affine else _distorted(mesh, 1 / 3, rng)
    elif kind == "QUAD4":
        mesh, okind, (w, p) = _distorted(fa.procedural.create_unit_square_uniform_quad_mesh_2d(4), 0.25, rng), oracle.QUAD4, quadrature.tensor.quadrilateral_gauss(n)
    else:
        mesh, okind, (w, p) = _distorted(fa.procedural.create_unit_box_uniform_tet_mesh_3d(2), 0.5, rng), oracle.TET4, quadrature.total_order.tetrahedron(n)
    asm, ref = _both(engine, oracle, mesh, okind, opname, w, p)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert st == 0
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert np.array_equal(k.col_indices, oci)
    assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max(), engine.last_kernel_name()


@pytest.mark.parametrize("kind", ["QUAD4", "HEX8", "TET4", "HEX27"])
def test_one_two_three_elements(engine, oracle, kind):
    rng = np.random.default_rng(5)
    if kind == "QUAD4":
        m0, (w, p), okind = fa.procedural.create_unit_square_uniform_quad_mesh_2d(2), quadrature.tensor.quadrilateral_gauss(2), oracle.QUAD4
    elif kind == "HEX8":
        m0, (w, p), okind = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 2), quadrature.tensor.hexahedron_gauss(2), oracle.HEX8
    elif kind == "TET4":
        m0, (w, p), okind = fa.procedural.create_unit_box_uniform_tet_mesh_3d(1), quadrature.total_order.tetrahedron(2), oracle.TET4
    else:
        m0 = fa.hex27_mesh_from_hex8(fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 2))
        (w, p), okind = quadrature.tensor.hexahedron_gauss(3), oracle.HEX27
    for ne, distort in itertools.product((1, 2, 3), (False, True)):
        conn = np.asarray(m0.connectivity)[:ne]
        used = np.unique(conn)
        remap = -np.ones(m0.num_nodes(), dtype=np.int64)
        remap[used] = np.arange(len(used))
        v = m0.vertices[used] + (rng.uniform(-0.05, 0.05, (len(used), m0.vertices.shape[1])) if distort else 0.0)
        mesh = fa.Mesh(v, remap[conn.astype(np.int64)].astype(np.uint64), m0.elem_kind)
        for opname in OPS:
            s = 1 if opname == "LAPLACE" else v.shape[1]
            asm, ref = _both(engine, oracle, mesh, okind, opname, w, p, u=1e-3 * rng.standard_normal(s * mesh.num_nodes()))
            st, _, oro, oci, ovals = oracle.assemble(ref)
            assert st == 0
            for scatter in (fa.SCATTER_GATHER, fa.SCATTER_ATOMIC):
                k = fa.CsrAssembler(scatter).assemble(asm)
                assert np.array_equal(k.row_offsets, oro) and np.array_equal(k.col_indices, oci)
                assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max(), (ne, distort, opname, engine.last_kernel_name())
            kc = fa.CsrParAssembler().assemble(fa.color_nodes(asm), asm)
            assert np.abs(kc.values - ovals).max() <= TOL * np.abs(ovals).max()
            f = fa.VectorAssembler().assemble_vector(asm)
            st, _, of = oracle.assemble_vector(ref)
            assert st == 0 and np.abs(f - of).max() <= 1e-11 * max(np.abs(of).max(), 1e-300)


@pytest.mark.parametrize("kind", ["HEX8", "TET4"])
@pytest.mark.parametrize("opname", ["LAPLACE", "LINEAR_ELASTIC"])
def test_permuted_numbering(engine, oracle, kind, opname):
    rng = np.random.default_rng(7)
    if kind == "HEX8":
        m0, (w, p), okind = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 5), quadrature.tensor.hexahedron_gauss(2), oracle.HEX8
    else:
        m0, (w, p), okind = fa.procedural.create_unit_box_uniform_tet_mesh_3d(3), quadrature.total_order.tetrahedron(1), oracle.TET4
    n = m0.num_nodes()
    perm = rng.permutation(n)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(n)
    conn = inv[np.asarray(m0.connectivity).astype(np.int64)][rng.permutation(m0.num_elements())]
    mesh = fa.Mesh(m0.vertices[perm], conn.astype(np.uint64), m0.elem_kind)
    asm, ref = _both(engine, oracle, mesh, okind, opname, w, p)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert st == 0 and np.array_equal(k.col_indices, oci)
    assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max(), engine.last_kernel_name()


@pytest.mark.parametrize("dims", [(20, 13, 9), (1, 1, 40), (40, 1, 1), (2, 37, 3), (31, 29, 17)])
@pytest.mark.parametrize("opname", ["LAPLACE", "LINEAR_ELASTIC"])
def test_graded_boxes_stay_on_the_affine_kernel(engine, oracle, dims, opname):
    """tensor-product meshes with a different spacing in every layer: every element is a box of its own size (affine, but no two records
    alike), the edge lengths are not multiples of the seven-node blocks, thin meshes have no interior node at all"""
    cx, cy, cz = dims
    m0 = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, cx, cy, cz, 1)
    v = m0.vertices.copy()
    for a, c in enumerate(dims):
        t = v[:, a] / c                                  # 0 .. 1
        v[:, a] = c * (0.35 * t + 0.65 * t ** 3) * (1.0 + 0.5 * a)   # monotone: boxes stay boxes
    mesh = fa.Mesh(v, m0.connectivity, m0.elem_kind)
    w, p = quadrature.tensor.hexahedron_gauss(2)
    asm, ref = _both(engine, oracle, mesh, oracle.HEX8, opname, w, p)
    st, _, oro, oci, ovals = oracle.assemble(ref)
    assert st == 0
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert "k_affine_rows" in engine.last_kernel_name()
    assert np.array_equal(k.col_indices, oci)
    assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max(), engine.last_kernel_name()
    # exact symmetry (util.rs:38-51) survives records that differ from element to element
    a = k.to_scipy()
    assert (a != a.T).nnz == 0


def test_randomised_gather_against_atomic():
    """scripts/fuzz_gather.py: 500 random small meshes (holes, permuted numbering, affine / distorted / mixed geometry, masks, row ranges,
    overwrite into garbage): the owner-computes kernels against the atomic scatter on the device"""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_gather

    bad, kernels = fuzz_gather.run(500, 20260, quiet=True)
    assert bad == 0
    assert sum(v for k, v in kernels.items() if "k_affine_rows" in k) > 20 and kernels.get("k_gather_rows", 0) > 50


def _scripts():
    import os
    import sys

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts")
    if path not in sys.path:
        sys.path.insert(0, path)


def test_randomised_vectors_and_energies():
    """scripts/fuzz_vector.py, 200 cases: the element pass (residual, energy: one thread per element) against the older LDS-staged kernels
    with a forced small grid, and the factored gravity source against sampled values -- all four operators, Hex8 / Tet4 / Quad4 / Tri3,
    holes, permutations, affine / distorted / mixed geometry, random rules"""
    _scripts()
    import fuzz_vector

    assert fuzz_vector.run(200, 4100, quiet=True) == 0


def test_randomised_patterns_against_scipy():
    """scripts/fuzz_pattern.py, 200 cases: assemble_pattern (global.rs:65-120) on random RAGGED connectivities (0 .. 40 nodes per element,
    repeated nodes, isolated nodes, hubs) for s = 1 .. 3 against the pattern of A^T A from scipy, bit for bit; the colouring valid"""
    _scripts()
    import fuzz_pattern

    assert fuzz_pattern.run(200, 7300, quiet=True) == 0
