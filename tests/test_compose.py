"""AggregateElementAssembler, MapElementNodes, TransformElement* (src/assembly/local.rs:152-340) -- the reference's own tests
(tests/unit_tests/assembly/local.rs:150-336) restated against the device path: repeated assembler, multibody block structure,
chained transformations; plus bodies of different element kinds and operators in one matrix, checked against the oracle."""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

pytestmark = pytest.mark.gpu
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))


def _laplace_quad(cells, rule=1):
    mesh = fa.procedural.create_unit_square_uniform_quad_mesh_2d(cells)
    w, p = quadrature.tensor.quadrilateral_gauss(rule)
    u = 0.1 * np.cos(np.arange(mesh.num_nodes()))
    asm = (fa.ElementEllipticAssemblerBuilder().with_operator(fa.LaplaceOperator()).with_finite_element_space(mesh)
           .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w)).with_u(u).build())
    return mesh, asm


def _dense(k):
    return k.to_scipy().toarray()


def test_map_element_nodes_connectivity():
    """local.rs tests: map_element_nodes changes populate_element_nodes and num_nodes, nothing else"""
    mesh, asm = _laplace_quad(2)
    mapped = asm.map_element_nodes(20, lambda i: 2 * i + 1)
    assert mapped.num_nodes() == 20 and mapped.num_elements() == asm.num_elements() and mapped.solution_dim() == 1
    a, b = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    for e in range(asm.num_elements()):
        asm.populate_element_nodes(a, e)
        mapped.populate_element_nodes(b, e)
        assert np.array_equal(b, 2 * a + 1)


def test_aggregate_element_assembler_repeated_assembler():
    """tests/unit_tests/assembly/local.rs:189-222"""
    _, asm = _laplace_quad(3)
    agg = fa.AggregateElementAssembler.from_assemblers([asm, asm])
    assert agg.num_elements() == 2 * asm.num_elements()
    assert abs(fa.assemble_scalar(agg) - 2.0 * fa.assemble_scalar(asm)) <= 1e-14 * abs(fa.assemble_scalar(asm))
    f, f1 = fa.VectorAssembler().assemble_vector(agg), fa.VectorAssembler().assemble_vector(asm)
    assert np.abs(f - 2.0 * f1).max() <= 1e-14 * np.abs(f1).max()
    k, k1 = fa.CsrAssembler().assemble(agg), fa.CsrAssembler().assemble(asm)
    assert np.array_equal(k.row_offsets, k1.row_offsets) and np.array_equal(k.col_indices, k1.col_indices)
    assert np.abs(k.values - 2.0 * k1.values).max() <= 1e-14 * np.abs(k1.values).max()


def test_aggregate_element_assembler_multibody():
    """tests/unit_tests/assembly/local.rs:224-290: [K1 0; 0 K2]"""
    m1, a1 = _laplace_quad(3)
    m2, a2 = _laplace_quad(4)
    n1, n = m1.num_nodes(), m1.num_nodes() + m2.num_nodes()
    agg = fa.AggregateElementAssembler.from_assemblers([a1.map_element_nodes(n, lambda i: i), a2.map_element_nodes(n, lambda i: i + n1)])
    assert agg.num_nodes() == n
    assert abs(fa.assemble_scalar(agg) - (fa.assemble_scalar(a1) + fa.assemble_scalar(a2))) <= 1e-13
    f = fa.VectorAssembler().assemble_vector(agg)
    assert len(f) == n
    assert np.array_equal(f[:n1], fa.VectorAssembler().assemble_vector(a1)) and np.array_equal(f[n1:], fa.VectorAssembler().assemble_vector(a2))
    K = _dense(fa.CsrAssembler().assemble(agg))
    assert np.array_equal(K[:n1, :n1], _dense(fa.CsrAssembler().assemble(a1)))
    assert np.array_equal(K[n1:, n1:], _dense(fa.CsrAssembler().assemble(a2)))
    assert not K[:n1, n1:].any() and not K[n1:, :n1].any()


def test_transform_element_scalar_vector_matrix():
    """tests/unit_tests/assembly/local.rs:292-336: layers of transformations combine in any order"""
    _, asm = _laplace_quad(3, rule=2)
    t = (asm.transform_element_scalar(-2.0).transform_element_vector(-1.5).transform_element_matrix(-3.0)
         .transform_element_scalar(2.0).transform_element_vector(2.0).transform_element_matrix(2.0))
    assert abs(fa.assemble_scalar(t) - (-4.0) * fa.assemble_scalar(asm)) <= 1e-13 * abs(fa.assemble_scalar(asm))
    f, f1 = fa.VectorAssembler().assemble_vector(t), fa.VectorAssembler().assemble_vector(asm)
    assert np.abs(f - (-3.0) * f1).max() <= 1e-14 * np.abs(f1).max()
    k, k1 = fa.CsrAssembler().assemble(t), fa.CsrAssembler().assemble(asm)
    assert np.abs(k.values - (-6.0) * k1.values).max() <= 1e-14 * np.abs(k1.values).max()


def test_two_bodies_of_different_element_kinds_share_nodes(oracle):
    """What the adapters are for: a Hex8 block (LinearElastic, gather kernels) and a Tet4 block (NeoHookean) glued along a
    face into ONE stiffness matrix over a common node space; against the oracle's sum over both bodies."""
    hexm = fa.procedural.create_unit_box_uniform_hex_mesh_3d(2)
    tetm = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
    # the tet box sits on top of the hex box (z + 1): the nodes of the hex top face z = 1 and of the tet bottom face z = 0
    # with equal (x, y) are identified
    tv = tetm.vertices + np.array([0.0, 0.0, 1.0])
    n_hex = hexm.num_nodes()
    gid = np.arange(n_hex, n_hex + tetm.num_nodes(), dtype=np.uint64)
    shared = 0
    for t, v in enumerate(tv):
        if abs(v[2] - 1.0) < 1e-12:
            hit = np.where(np.abs(hexm.vertices - v).max(axis=1) < 1e-12)[0]
            if len(hit):
                gid[t] = hit[0]
                shared += 1
    assert shared == 9
    # compact the global numbering
    used = np.unique(np.concatenate([np.arange(n_hex, dtype=np.uint64), gid]))
    remap = {int(g): i for i, g in enumerate(used)}
    gmap_tet = np.array([remap[int(g)] for g in gid], dtype=np.uint64)
    n = len(used)
    rng = np.random.default_rng(4)
    ug = 0.01 * rng.standard_normal(3 * n)
    wh, ph = quadrature.tensor.hexahedron_gauss(2)
    wt, pt = quadrature.total_order.tetrahedron(2)
    u_hex = ug.reshape(-1, 3)[:n_hex].reshape(-1)
    u_tet = ug.reshape(-1, 3)[gmap_tet.astype(int)].reshape(-1)
    a_hex = (fa.ElementEllipticAssemblerBuilder().with_finite_element_space(hexm)
             .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
             .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(ph, wh).with_uniform_data(LAME)).with_u(u_hex).build())
    a_tet = (fa.ElementEllipticAssemblerBuilder().with_finite_element_space(fa.Mesh(tv, tetm.connectivity, fa.TET4))
             .with_operator(fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()))
             .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(pt, wt).with_uniform_data(LAME)).with_u(u_tet).build())
    agg = fa.AggregateElementAssembler.from_assemblers([a_hex.map_element_nodes(n, np.arange(n_hex, dtype=np.uint64)),
                                                        a_tet.map_element_nodes(n, gmap_tet)])
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(agg)
    # oracle: both bodies in the global numbering, accumulated into the aggregate pattern
    gverts = np.zeros((n, 3))
    gverts[:n_hex] = hexm.vertices
    gverts[gmap_tet.astype(int)] = tv
    o_hex = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, gverts, hexm.connectivity, wh, ph, params=LAME.as_pair(), u=ug)
    o_tet = oracle.ElementAssembler(oracle.TET4, oracle.NEO_HOOKEAN, gverts, gmap_tet[tetm.connectivity.astype(int)], wt, pt,
                                    params=LAME.as_pair(), u=ug)
    vals = np.zeros(len(k.col_indices))
    for o in (o_hex, o_tet):
        st, _ = oracle.assemble_into_csr(o, k.row_offsets, k.col_indices, vals)
        assert st == 0
    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
    f = fa.VectorAssembler().assemble_vector(agg)
    fo = np.zeros(3 * n)
    for o in (o_hex, o_tet):
        st, _, fo = oracle.assemble_vector(o, out=fo)
    assert np.abs(f - fo).max() <= 1e-12 * np.abs(fo).max()


def test_mapped_add_reports_missing_entries():
    """a destination pattern that lacks a mapped entry: FH_BAD_ARGUMENT (the reference panics, global.rs:531-533)"""
    import ctypes as C

    import torch

    from fenris_amd import _ffi

    _, a1 = _laplace_quad(2)
    _, a2 = _laplace_quad(3)
    eng = a1.engine
    nnz = eng.build_pattern()
    part = torch.ones(nnz, dtype=torch.float64, device="cuda")
    ro, ci = a2.engine.pattern()
    ro_t, ci_t = torch.from_numpy(ro.view(np.int64)).cuda(), torch.from_numpy(ci.view(np.int64)).cuda()
    dst = torch.zeros(len(ci), dtype=torch.float64, device="cuda")
    # identity map of a 3 x 3-node mesh into the pattern of a 4 x 4-node mesh: node 2 and node 3 are not neighbours there
    rc = _ffi.lib().fh_add_mapped_matrix_dev(eng._h, C.c_void_p(part.data_ptr()), None, 1.0, a2.num_nodes(), C.c_void_p(ro_t.data_ptr()),
                                             C.c_void_p(ci_t.data_ptr()), C.c_void_p(dst.data_ptr()))
    assert rc == _ffi.FH_BAD_ARGUMENT


def test_aggregate_and_mapped_source_assemblers():
    """The multi-body right-hand side (local.rs:152-340 over ElementSourceAssembler bodies, local/source.rs:159-278): two bodies
    with their own meshes and gravity sources mapped into one node space, one of them scaled -- the bodies' contexts hold no
    operator, the solution dimension goes along explicitly (fh_add_mapped_vector_sdim_dev)."""
    def body(cells, g, rho0):
        m = fa.procedural.create_unit_square_uniform_quad_mesh_2d(cells)
        w, p = quadrature.tensor.quadrilateral_gauss(2)
        rho = np.linspace(rho0, rho0 + 1.0, len(w))
        qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_data([fa.Density(r) for r in rho])
        asm = (fa.ElementSourceAssemblerBuilder.new(fa.Engine(0)).with_finite_element_space(m)
               .with_source(fa.GravitySource.from_acceleration(np.asarray(g))).with_quadrature_table(qt).build())
        return m, asm

    m1, s1 = body(3, [0.0, -9.81], 1.0)
    m2, s2 = body(4, [1.5, -2.0], 3.0)
    f1, f2 = fa.VectorAssembler().assemble_vector(s1), fa.VectorAssembler().assemble_vector(s2)
    n1, n = m1.num_nodes(), m1.num_nodes() + m2.num_nodes()
    agg = fa.AggregateElementAssembler.from_assemblers([s1.map_element_nodes(n, lambda i: i),
                                                        s2.map_element_nodes(n, lambda i: i + n1).transform_element_vector(-0.5)])
    assert agg.solution_dim() == 2 and agg.num_nodes() == n
    f = fa.VectorAssembler().assemble_vector(agg)
    ref = np.concatenate([f1, -0.5 * f2])
    assert np.abs(f - ref).max() <= 1e-14 * np.abs(ref).max()
    # the same body twice: twice the vector
    twice = fa.AggregateElementAssembler.from_assemblers([s1, s1])
    assert np.abs(fa.VectorAssembler().assemble_vector(twice) - 2.0 * f1).max() <= 1e-14 * np.abs(f1).max()
