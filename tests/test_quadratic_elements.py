"""Tet10 / Quad9 / Tri6 / Hex20 / Tet20 (sub-parametric higher-order elements): oracle pins on the CPU, HIP parity and the reference's
MMS error JSONs on the GPU."""
import json
import os

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature
from conftest import GOLDEN

KIND = {"TET10": fa.TET10, "QUAD9": fa.QUAD9, "TRI6": fa.TRI6, "HEX20": fa.HEX20, "TET20": fa.TET20}
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
OPS = {"LAPLACE": lambda: fa.LaplaceOperator(), "LINEAR_ELASTIC": lambda: fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
       "NEO_HOOKEAN": lambda: fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()), "STVK": lambda: fa.MaterialEllipticOperator(fa.StVKMaterial())}
T = 1.0 / 3.0
REF_NODES = {  # reference elements: tetrahedron.rs:153-168, quadrilateral.rs:212-228, triangle.rs:196-206
    "TET10": [[-1, -1, -1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1], [0, -1, -1], [0, 0, -1], [-1, 0, -1], [-1, -1, 0], [-1, 0, 0], [0, -1, 0]],
    "QUAD9": [[-1, -1], [1, -1], [1, 1], [-1, 1], [0, -1], [1, 0], [0, 1], [-1, 0], [0, 0]],
    "TRI6": [[-1, -1], [1, -1], [-1, 1], [0, -1], [0, 0], [-1, 0]],
    # hexahedron.rs:376-402: corners, then the edge nodes in the order of Hex27
    "HEX20": [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1],
              [0, -1, -1], [-1, 0, -1], [-1, -1, 0], [1, 0, -1], [1, -1, 0], [0, 1, -1], [1, 1, 0], [-1, 1, 0],
              [0, -1, 1], [-1, 0, 1], [1, 0, 1], [0, 1, 1]],
    # tetrahedron.rs:296-340: vertices, two nodes per edge (01, 02, 03, 12, 13, 23), face nodes (012, 013, 023, 123)
    "TET20": [[-1, -1, -1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1],
              [-T, -1, -1], [T, -1, -1], [-1, -T, -1], [-1, T, -1], [-1, -1, -T], [-1, -1, T],
              [T, -T, -1], [-T, T, -1], [T, -1, -T], [-T, -1, T], [-1, T, -T], [-1, -T, T],
              [-T, -T, -1], [-T, -1, -T], [-1, -T, -T], [-T, -T, -T]],
}


def _tri_rule6():
    t = json.load(open(os.path.join(GOLDEN, "tri_rule_6_12.json")))
    return np.array(t["weights"]), np.array(t["points"])


def _tet_rule6():
    t = json.load(open(os.path.join(GOLDEN, "tet_rule_6_24.json")))
    return np.array(t["weights"]), np.array(t["points"])


def _mesh(kind, seed=0, distort=0.04):
    rng = np.random.default_rng(seed)
    if kind == "TET10":
        base = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
    elif kind == "QUAD9":
        base = fa.procedural.create_unit_square_uniform_quad_mesh_2d(3)
    elif kind == "HEX20":
        base = fa.procedural.create_unit_box_uniform_hex_mesh_3d(2)
    elif kind == "TET20":
        base = fa.procedural.create_unit_box_uniform_tet_mesh_3d(1)
    else:
        base = fa.procedural.create_unit_square_uniform_tri_mesh_2d(3)
    base = fa.Mesh(base.vertices + rng.uniform(-distort, distort, base.vertices.shape), base.connectivity, base.elem_kind)
    return {"TET10": fa.tet10_mesh_from_tet4, "QUAD9": fa.quad9_mesh_from_quad4, "TRI6": fa.tri6_mesh_from_tri3,
            "HEX20": fa.hex20_mesh_from_hex8, "TET20": fa.tet20_mesh_from_tet4}[kind](base), base


def _rule(kind):
    if kind == "TET10":
        return quadrature.total_order.tetrahedron(3)
    if kind == "TET20":
        return quadrature.total_order.tetrahedron(5)
    if kind == "QUAD9":
        return quadrature.tensor.quadrilateral_gauss(3)
    if kind == "HEX20":
        return quadrature.tensor.hexahedron_gauss(3)
    return _tri_rule6()


# ------------------------------------------------------------------------------------------- CPU: oracle pins
@pytest.mark.parametrize("kind", ["TET10", "QUAD9", "TRI6", "HEX20", "TET20"])
def test_oracle_basis_is_nodal_and_a_partition_of_unity(oracle, kind):
    """the properties the reference's element tests check (partition of unity, Lagrange property at the reference
    nodes, gradients consistent with finite differences)"""
    k = getattr(oracle, kind)
    nodes = np.array(REF_NODES[kind], dtype=float)
    n = len(nodes)
    vals = np.array([oracle.element_basis(k, x) for x in nodes])
    assert np.abs(vals - np.eye(n)).max() < 1e-14
    rng = np.random.default_rng(0)
    for xi in rng.uniform(-0.9, 0.2, (5, nodes.shape[1])):
        phi = oracle.element_basis(k, xi)
        g = oracle.element_gradients(k, xi).T  # (n, d)
        assert abs(phi.sum() - 1.0) < 1e-14 and np.abs(g.sum(axis=0)).max() < 1e-13
        h = 1e-6
        for c in range(nodes.shape[1]):
            e = np.zeros(nodes.shape[1]); e[c] = h
            fd = (oracle.element_basis(k, xi + e) - oracle.element_basis(k, xi - e)) / (2 * h)
            assert np.abs(fd - g[:, c]).max() < 1e-8


def test_tet20_conversion_product_equals_oracle_and_is_consistent(oracle):
    base = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
    rng = np.random.default_rng(5)
    base = fa.Mesh(base.vertices + rng.uniform(-0.03, 0.03, base.vertices.shape), base.connectivity, base.elem_kind)
    mesh = fa.tet20_mesh_from_tet4(base)
    ov, oc = oracle.tet4_to_tet20(base.vertices, base.connectivity)
    assert np.array_equal(mesh.vertices, ov) and np.array_equal(mesh.connectivity, oc)
    c, b = mesh.connectivity.astype(int), base.connectivity.astype(int)
    assert np.array_equal(mesh.vertices[c[:, :4]], base.vertices[b])  # the vertex nodes are the Tet4 vertices
    edges = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    for m, (i, j) in enumerate(edges):  # thirds of the edges, the first node next to vertex i
        vi, vj = mesh.vertices[c[:, i]], mesh.vertices[c[:, j]]
        assert np.abs(mesh.vertices[c[:, 4 + 2 * m]] - (vi + (vj - vi) / 3)).max() < 1e-14
        assert np.abs(mesh.vertices[c[:, 5 + 2 * m]] - (vi + 2 * (vj - vi) / 3)).max() < 1e-14
    for f, tri in enumerate([(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)]):
        cen = sum(mesh.vertices[c[:, k]] for k in tri) / 3
        assert np.abs(mesh.vertices[c[:, 16 + f]] - cen).max() < 1e-14
    n_edges = len({tuple(sorted((r[i], r[j]))) for r in b for i, j in edges})
    n_faces = len({tuple(sorted((r[i], r[j], r[k]))) for r in b for i, j, k in [(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)]})
    assert mesh.num_nodes() == base.num_nodes() + 2 * n_edges + n_faces


@pytest.mark.parametrize("kind", ["TET10", "QUAD9", "TRI6", "HEX20"])
def test_refinement_product_equals_oracle_and_is_consistent(oracle, kind):
    mesh, base = _mesh(kind, seed=1)
    ov, oc = oracle.refine_to_quadratic(getattr(oracle, {"TET10": "TET4", "QUAD9": "QUAD4", "TRI6": "TRI3", "HEX20": "HEX8"}[kind]), base.vertices,
                                        base.connectivity)
    assert np.array_equal(mesh.vertices, ov) and np.array_equal(mesh.connectivity, oc)  # bit-exact
    # edge nodes are the midpoints of their end vertices, shared edges share their node
    c = mesh.connectivity.astype(int)
    edges = {"TET10": [(0, 1), (1, 2), (0, 2), (0, 3), (2, 3), (1, 3)], "QUAD9": [(0, 1), (1, 2), (2, 3), (3, 0)],
             "TRI6": [(0, 1), (1, 2), (2, 0)],
             "HEX20": [(0, 1), (0, 3), (0, 4), (1, 2), (1, 5), (2, 3), (2, 6), (3, 7), (4, 5), (4, 7), (5, 6), (6, 7)]}[kind]
    nv = len(edges) + (0 if kind == "QUAD9" else 0)
    first = {"TET10": 4, "QUAD9": 4, "TRI6": 3, "HEX20": 8}[kind]
    for m, (a, b) in enumerate(edges):
        mid = 0.5 * (mesh.vertices[c[:, a]] + mesh.vertices[c[:, b]])
        assert np.abs(mesh.vertices[c[:, first + m]] - mid).max() < 1e-15
    n_edges = len({tuple(sorted((row[a], row[b]))) for row in base.connectivity.astype(int) for a, b in edges})
    expect = base.num_nodes() + n_edges + (base.num_elements() if kind == "QUAD9" else 0)
    assert mesh.num_nodes() == expect


@pytest.mark.parametrize("name,kind,nres", [("poisson2d_mms_quad9_summary", "QUAD9", 3), ("poisson2d_mms_tri6_summary", "TRI6", 3),
                                            ("poisson3d_mms_tet10_summary", "TET10", 2), ("poisson3d_mms_hex20_summary", "HEX20", 2),
                                            ("poisson3d_mms_tet20_summary", "TET20", 2)])
def test_oracle_mms_errors(oracle, name, kind, nres):
    """tests/convergence_tests/poisson_{2d,3d}_mms.rs with the oracle end to end (assembly, source, Dirichlet, CG, error
    norms) against the reference's error JSONs (1 %)"""
    _mms(name, kind, nres, None, oracle)


def _mms(name, kind, nres, engines, oracle):
    ref = json.load(open(os.path.join(GOLDEN, "mms_reference_values.json")))["summaries"][name]
    if kind == "QUAD9":
        gen = lambda r: fa.quad9_mesh_from_quad4(fa.procedural.create_unit_square_uniform_quad_mesh_2d(r))
        rule, err_rule = quadrature.tensor.quadrilateral_gauss(2), quadrature.tensor.quadrilateral_gauss(6)
    elif kind == "TRI6":
        gen = lambda r: fa.tri6_mesh_from_tri3(fa.procedural.create_unit_square_uniform_tri_mesh_2d(r))
        rule, err_rule = quadrature.total_order.triangle(2), _tri_rule6()
    elif kind == "TET20":  # poisson_3d_mms.rs:131-138 (resolutions 1, 2, 4, 6, ...)
        gen = lambda r: fa.tet20_mesh_from_tet4(fa.procedural.create_unit_box_uniform_tet_mesh_3d(r))
        rule, err_rule = quadrature.total_order.tetrahedron(4), _tet_rule6()
    elif kind == "HEX20":  # poisson_3d_mms.rs:91-98
        gen = lambda r: fa.hex20_mesh_from_hex8(fa.procedural.create_unit_box_uniform_hex_mesh_3d(r))
        rule, err_rule = quadrature.tensor.hexahedron_gauss(4), quadrature.tensor.hexahedron_gauss(6)
    else:
        gen = lambda r: fa.tet10_mesh_from_tet4(fa.procedural.create_unit_box_uniform_tet_mesh_3d(r))
        rule, err_rule = quadrature.total_order.tetrahedron(2), _tet_rule6()

    def u_exact(x):
        return np.prod(np.sin(np.pi * x), axis=-1)[..., None]

    def u_grad(x):
        d = x.shape[-1]
        g = np.zeros(x.shape[:-1] + (d, 1))
        for i in range(d):
            t = np.pi * np.cos(np.pi * x[..., i])
            for j in range(d):
                if j != i:
                    t = t * np.sin(np.pi * x[..., j])
            g[..., i, 0] = t
        return g

    for i, res in enumerate([1, 2, 4, 8, 16][:nres]):
        mesh = gen(res)
        w, p = rule
        we, pe = err_rule
        d, N = mesh.vertices.shape[1], mesh.num_nodes()
        bc = np.where(np.abs(mesh.vertices - 0.5).max(axis=1) > 0.4999)[0]
        if engines is None:
            okind = getattr(oracle, kind)
            asm = oracle.ElementAssembler(okind, oracle.LAPLACE, mesh.vertices, mesh.connectivity, w, p)
            st, _, ro, ci, vals = oracle.assemble(asm)
            assert st == 0
            xq = oracle.physical_quadrature_points(asm)
            st, b = oracle.assemble_source_vector(asm, 1, values=d * np.pi ** 2 * u_exact(xq))
            oracle.apply_homogeneous_dirichlet_bc_csr(ro, ci, vals, bc, 1)
            b[bc] = 0.0
            st, uh, _ = oracle.cg_solve(ro, ci, vals, b, jacobi=True, tol=1e-9, max_iter=10000)
            assert st == 0
            easm = oracle.ElementAssembler(okind, oracle.LAPLACE, mesh.vertices, mesh.connectivity, we, pe)
            xe = oracle.physical_quadrature_points(easm)
            l2 = np.sqrt(oracle.estimate_error_squared(easm, 0, 1, uh, u_exact(xe))[1])
            h1 = np.sqrt(oracle.estimate_error_squared(easm, 1, 1, uh, u_grad(xe))[1])
        else:
            import torch

            e_k, e_b, e_err = engines
            qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
            lap = (fa.ElementEllipticAssemblerBuilder(e_k).with_finite_element_space(mesh).with_operator(fa.LaplaceOperator())
                   .with_quadrature_table(qt).with_u(np.zeros(N)).build())
            k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(lap, device_values=True)
            src = (fa.ElementSourceAssemblerBuilder.new(e_b).with_finite_element_space(mesh)
                   .with_source(fa.SourceFunction(1, lambda x, _d: d * np.pi ** 2 * u_exact(x))).with_quadrature_table(qt).build())
            b = torch.zeros(N, dtype=torch.float64, device="cuda:0")
            fa.VectorAssembler().assemble_vector_into(b, src)
            fa.apply_homogeneous_dirichlet_bc_csr(k, bc, 1, lap)
            fa.apply_homogeneous_dirichlet_bc_rhs(b, bc, 1)
            u_h = torch.zeros(N, dtype=torch.float64, device="cuda:0")
            (fa.ConjugateGradient.new().with_operator(k, lap).with_preconditioner(fa.JacobiPreconditioner()).with_max_iter(10000)
             .with_stopping_criterion(fa.RelativeResidualCriterion(1e-9)).solve_with_guess(b, u_h))
            err_asm = (fa.ElementSourceAssemblerBuilder.new(e_err).with_finite_element_space(mesh)
                       .with_source(fa.SourceFunction(1, lambda x, _d: u_exact(x)))
                       .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(pe, we)).build())
            uh = u_h.cpu().numpy()
            l2 = fa.estimate_L2_error(err_asm, u_exact, uh)
            h1 = fa.estimate_H1_seminorm_error(err_asm, u_grad, uh)
        assert abs(l2 - ref["L2_errors"][i]) / ref["L2_errors"][i] < 0.01, (res, l2, ref["L2_errors"][i])
        assert abs(h1 - ref["H1_seminorm_errors"][i]) / ref["H1_seminorm_errors"][i] < 0.01, (res, h1)


# ------------------------------------------------------------------------------------------- GPU parity
@pytest.fixture(scope="module")
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["TET10", "QUAD9", "TRI6", "HEX20", "TET20"])
@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC", "NEO_HOOKEAN", "STVK"])
def test_matrix_vector_scalar_match_oracle(engine, oracle, kind, op):
    mesh, _ = _mesh(kind, seed=2)
    w, p = _rule(kind)
    d = mesh.vertices.shape[1]
    s = 1 if op == "LAPLACE" else d
    u = 0.01 * np.random.default_rng(3).standard_normal(s * mesh.num_nodes())
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    params = None
    if op != "LAPLACE":
        qt, params = qt.with_uniform_data(LAME), np.array(LAME.as_pair())
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(OPS[op]())
           .with_quadrature_table(qt).with_u(u).build())
    ref = oracle.ElementAssembler(getattr(oracle, kind), getattr(oracle, op), mesh.vertices, mesh.connectivity, w, p, params=params, u=u)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    for scatter in (fa.SCATTER_GATHER, fa.SCATTER_ATOMIC):
        k = fa.CsrAssembler(scatter).assemble(asm)
        assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
        assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max(), (scatter, engine.last_kernel_name())
    # the gather runs as two passes on these elements; between them the element matrices are lower node-block triangles where the blocks are 3 x 3
    # (3D elasticity family) and full column-major matrices otherwise (and under FENRIS_HIP_TWO_PASS_FULL): same matrix to rounding, and the
    # triangle form adds the same stored doubles to (I, J) and (J, I): symmetric bit for bit
    kg = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    tri = op != "LAPLACE" and d == 3
    two_pass = engine.last_kernel_name().startswith("k_assemble_matrix<dump>")     # (Tri6 with a linear operator: six nodes, one pass)
    assert two_pass or kind == "TRI6"
    if two_pass:
        assert engine.last_kernel_name() == "k_assemble_matrix<dump> + " + ("k_rows_from_tri" if tri else "k_rows_from_dense")
    if tri:
        a = kg.to_scipy()
        dd = (a - a.T).tocoo()
        assert dd.nnz == 0 or not np.any(dd.data != 0.0)
        engine.set_option("FENRIS_HIP_TWO_PASS_FULL", 1)
        try:
            kf = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
            assert engine.last_kernel_name() == "k_assemble_matrix<dump> + k_rows_from_dense"
        finally:
            engine.set_option("FENRIS_HIP_TWO_PASS_FULL", None)
        assert np.abs(kf.values - kg.values).max() <= 1e-13 * np.abs(vals).max()
        assert np.array_equal(fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm).values, kg.values)
    k = fa.CsrParAssembler().assemble(fa.color_nodes(asm), asm)
    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
    f = fa.VectorAssembler().assemble_vector(asm)
    st, _, fo = oracle.assemble_vector(ref)
    assert st == 0 and np.abs(f - fo).max() <= 1e-12 * max(np.abs(fo).max(), 1e-300)
    e = fa.assemble_scalar(asm)
    st, _, eo = oracle.assemble_scalar(ref)
    assert st == 0 and abs(e - eo) <= 1e-12 * max(abs(eo), 1e-300)
    ke = engine.element_matrices(0, min(3, mesh.num_elements()))
    for el in range(len(ke)):
        st, ko = ref.element_matrix(el)
        assert st == 0 and np.abs(ke[el] - ko).max() <= 1e-12 * np.abs(ko).max()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["TET10", "QUAD9", "TRI6", "HEX20", "TET20"])
def test_mass_and_source_match_oracle(engine, oracle, kind):
    mesh, _ = _mesh(kind, seed=4)
    w, p = _rule(kind)
    d = mesh.vertices.shape[1]
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.Density(2.5))
    mass = fa.ElementMassAssembler.with_solution_dim(d, engine).with_space(mesh).with_quadrature_table(qt)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(mass)
    oasm = oracle.ElementAssembler(getattr(oracle, kind), oracle.MASS_VECTOR, mesh.vertices, mesh.connectivity, w, p, params=[2.5, 0.0])
    vals = oracle.assemble(oasm)[4]
    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
    g = np.array([0.3, -9.81, 1.2])[:d]
    src = (fa.ElementSourceAssemblerBuilder.new(engine).with_finite_element_space(mesh).with_source(fa.GravitySource(g))
           .with_quadrature_table(qt).build())
    f = fa.VectorAssembler().assemble_vector(src)
    st, fo = oracle.assemble_source_vector(oasm, d, g=g)
    assert st == 0 and np.abs(f - fo).max() <= 1e-12 * np.abs(fo).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,nres", [("poisson2d_mms_quad9_summary", "QUAD9", 4), ("poisson2d_mms_tri6_summary", "TRI6", 4),
                                            ("poisson3d_mms_tet10_summary", "TET10", 3), ("poisson3d_mms_hex20_summary", "HEX20", 3),
                                            ("poisson3d_mms_tet20_summary", "TET20", 3)])
def test_mms_loop_on_device(oracle, name, kind, nres):
    engines = (fa.Engine(0), fa.Engine(0), fa.Engine(0))
    try:
        _mms(name, kind, nres, engines, oracle)
    finally:
        for e in engines:
            e.close()
