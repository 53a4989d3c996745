"""Residual and source vectors through element tiles (fenris_amd/csrc/vector_tiles.hip) against the oracle: what replaces
VectorAssembler::assemble_vector_into (global.rs:582-608) with assemble_element_elliptic_vector (local/elliptic.rs:457-531) and
ElementSourceAssembler (local/source.rs:159-278) for Quad4 / Tri3 / Tet4 / Hex8.

The tables are built on the device from the connectivity (Morton order of the centroids -> tiles of 256 elements -> distinct nodes per tile ->
node -> partials); the cases below aim at their corners: several tiles with a ragged last one, permuted numbering (a tile's nodes scattered
over the id range), element soups (every element its own nodes: 2048 distinct nodes in a Hex8 tile, the table's upper bound), nodes
without elements, coincident centroids, an element mask (the multi-GPU partitions: inactive elements contribute nothing, also when they
hold NaN), accumulation into the output, a second mesh on the same context, bitwise reproducibility."""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

pytestmark = pytest.mark.gpu
LAME = fa.LameParameters(3.0e2, 5.0e2)
TILED = "k_element_pass_tiled + k_vector_from_partials"


def _permuted(mesh, seed):
    rng = np.random.default_rng(seed)
    vp = rng.permutation(mesh.num_nodes())
    inv = np.empty_like(vp)
    inv[vp] = np.arange(len(vp))
    conn = inv[np.asarray(mesh.connectivity).astype(np.int64)][rng.permutation(mesh.num_elements())]
    return fa.Mesh(mesh.vertices[vp], conn.astype(np.uint64), mesh.elem_kind)


def _soup(mesh):
    """every element gets its own copies of its vertices (+ two unused vertices in front)"""
    conn = np.asarray(mesh.connectivity).astype(np.int64)
    verts = np.concatenate([np.zeros((2, mesh.vertices.shape[1])), mesh.vertices[conn.reshape(-1)]])
    return fa.Mesh(verts, (2 + np.arange(conn.size).reshape(conn.shape)).astype(np.uint64), mesh.elem_kind)


def _cases(oracle):
    rng = np.random.default_rng(11)
    h = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 9, 7, 6, 1)                     # 378 elements: two tiles, the second ragged
    hv = fa.Mesh(h.vertices + 0.02 * rng.uniform(-1, 1, h.vertices.shape), h.connectivity, fa.HEX8)
    yield "hex8 distorted", hv, oracle.HEX8, quadrature.tensor.hexahedron_gauss(2)
    yield "hex8 permuted", _permuted(hv, 1), oracle.HEX8, quadrature.tensor.hexahedron_gauss(2)
    yield "hex8 soup", _soup(fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 8, 8, 5, 1)), oracle.HEX8, quadrature.tensor.hexahedron_gauss(2)
    t = fa.procedural.create_unit_box_uniform_tet_mesh_3d(4)                                   # 768 tetrahedra: three tiles
    yield "tet4 permuted", _permuted(t, 2), oracle.TET4, quadrature.total_order.tetrahedron(2)
    yield "tet4 soup", _soup(t), oracle.TET4, quadrature.total_order.tetrahedron(1)
    q = fa.procedural.create_unit_square_uniform_quad_mesh_2d(20)
    yield "quad4 permuted", _permuted(fa.Mesh(q.vertices + 0.004 * rng.uniform(-1, 1, q.vertices.shape), q.connectivity, fa.QUAD4), 3), oracle.QUAD4, \
        quadrature.tensor.quadrilateral_gauss(2)
    flat = fa.Mesh(np.zeros_like(hv.vertices), hv.connectivity, fa.HEX8)                        # all centroids coincide, every element singular
    yield "hex8 collapsed", flat, oracle.HEX8, quadrature.tensor.hexahedron_gauss(2)


def _assembler(engine, mesh, opname, w, p, u):
    ops = {"LAPLACE": fa.LaplaceOperator, "LINEAR_ELASTIC": lambda: fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
           "NEO_HOOKEAN": lambda: fa.MaterialEllipticOperator(fa.NeoHookeanMaterial())}
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if opname != "LAPLACE":
        qt = qt.with_uniform_data(LAME)
    return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(ops[opname]()).with_quadrature_table(qt)
            .with_u(u).build())


def _oracle_vector(oracle, okind, opname, mesh, conn, w, p, u):
    ref = oracle.ElementAssembler(okind, getattr(oracle, opname), mesh.vertices, conn, w, p, params=(LAME.as_pair() if opname != "LAPLACE" else None), u=u)
    st, _, f = oracle.assemble_vector(ref)
    return st, f


def test_tiled_residual_matches_the_oracle_with_and_without_mask(oracle):
    import torch

    rng = np.random.default_rng(5)
    engine = fa.Engine(0)          # ONE context walks all meshes: the tables follow the topology
    try:
        for name, mesh, okind, (w, p) in _cases(oracle):
            d = mesh.vertices.shape[1]
            for opname in ("LAPLACE", "LINEAR_ELASTIC", "NEO_HOOKEAN"):
                if name == "hex8 collapsed" and opname != "LINEAR_ELASTIC":
                    continue
                s = 1 if opname == "LAPLACE" else d
                u = 0.004 * rng.standard_normal(s * mesh.num_nodes())
                asm = _assembler(engine, mesh, opname, w, p, u)
                st, want = _oracle_vector(oracle, okind, opname, mesh, mesh.connectivity, w, p, u)
                if name == "hex8 collapsed":
                    assert st == 1                                   # singular Jacobian: the error of the reference, through the tiled kernel
                    with pytest.raises(Exception):
                        fa.VectorAssembler().assemble_vector(asm)
                    assert engine.last_kernel_name() == TILED
                    continue
                assert st == 0
                got = fa.VectorAssembler().assemble_vector(asm)
                assert engine.last_kernel_name() == TILED, (name, opname)
                assert np.all(np.isfinite(want)), (name, opname)
                scale = max(np.abs(want).max(), 1e-300)
                assert np.abs(got - want).max() <= 1e-12 * scale, (name, opname)
                # accumulation into the output, device resident; the same bits every time
                out = torch.full((s * mesh.num_nodes(),), 2.0, dtype=torch.float64, device="cuda:0")
                engine.assemble_vector(out)
                first = out.cpu().numpy().copy()
                assert np.abs(first - 2.0 - want).max() <= 1e-12 * max(scale, 2.0)
                out.fill_(2.0)
                engine.assemble_vector(out)
                assert np.array_equal(out.cpu().numpy(), first), (name, opname)
                # element mask: the inactive elements contribute nothing -- also when they hold NaN (NeoHookean: a large u on their nodes only)
                mask = (rng.random(mesh.num_elements()) < 0.6).astype(np.uint8)
                conn = np.asarray(mesh.connectivity)
                um = u.copy()
                if opname == "NEO_HOOKEAN":
                    only_inactive = np.setdiff1d(conn[mask == 0].reshape(-1), conn[mask == 1].reshape(-1))
                    um.reshape(-1, s)[only_inactive] = 50.0 * rng.standard_normal((len(only_inactive), s))
                    asm = _assembler(engine, mesh, opname, w, p, um)
                engine.set_active_elements(mask)
                st, want_m = _oracle_vector(oracle, okind, opname, mesh, conn[mask == 1], w, p, um)
                assert st == 0
                got_m = fa.VectorAssembler().assemble_vector(asm)
                assert engine.last_kernel_name() == TILED
                assert np.all(np.isfinite(got_m)), (name, opname)
                assert np.abs(got_m - want_m).max() <= 1e-12 * max(np.abs(want_m).max(), 1e-300), (name, opname, "masked")
                # the energy over the same tiles: the active elements only (the inactive ones may hold inf / NaN)
                sub = oracle.ElementAssembler(okind, getattr(oracle, opname), mesh.vertices, conn[mask == 1], w, p,
                                              params=(LAME.as_pair() if opname != "LAPLACE" else None), u=um)
                res = oracle.assemble_scalar(sub)
                assert res[0] == 0
                e_gpu = fa.assemble_scalar(asm)
                assert engine.last_kernel_name() == "k_element_energy_tiled"
                assert abs(e_gpu - res[-1]) <= 1e-12 * max(abs(res[-1]), 1e-300), (name, opname, "masked energy")
                engine.set_active_elements(None)
                e_all = fa.assemble_scalar(_assembler(engine, mesh, opname, w, p, u))
                ref_all = oracle.assemble_scalar(oracle.ElementAssembler(okind, getattr(oracle, opname), mesh.vertices, mesh.connectivity, w, p,
                                                                         params=(LAME.as_pair() if opname != "LAPLACE" else None), u=u))
                assert abs(e_all - ref_all[-1]) <= 1e-12 * max(abs(ref_all[-1]), 1e-300), (name, opname, "energy")
    finally:
        engine.close()


@pytest.mark.parametrize("sdim", [1, 3])
def test_tiled_source_vector_matches_the_oracle(oracle, sdim):
    rng = np.random.default_rng(8)
    g = np.array([0.3, -9.81, 1.7])[:sdim]
    engine = fa.Engine(0)
    try:
        for name, mesh, okind, (w, p) in _cases(oracle):
            if mesh.vertices.shape[1] != 3 or name == "hex8 collapsed":
                continue
            rho = 1.0 + rng.random(len(w))
            qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_data([fa.Density(r) for r in rho])
            asm = (fa.ElementSourceAssemblerBuilder.new(engine).with_finite_element_space(mesh).with_source(fa.GravitySource.from_acceleration(g))
                   .with_quadrature_table(qt).build())
            oasm = oracle.ElementAssembler(okind, oracle.LAPLACE, mesh.vertices, mesh.connectivity, w, p, params=np.stack([rho, np.zeros_like(rho)], axis=1))
            st, want = oracle.assemble_source_vector(oasm, sdim, g=g)
            assert st == 0
            got = fa.VectorAssembler().assemble_vector(asm)
            assert engine.last_kernel_name() == "k_source_elements_tiled + k_vector_from_partials", name
            assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), name
            assert np.array_equal(fa.VectorAssembler().assemble_vector(asm), got)              # the same bits every time
            # sampled values per (element, point): the unfactored form
            vals = rng.standard_normal((mesh.num_elements(), len(w), sdim))
            asm2 = (fa.ElementSourceAssemblerBuilder.new(engine).with_finite_element_space(mesh)
                    .with_source(fa.SourceFunction(sdim, lambda x, _d, v=vals: v)).with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w)).build())
            st, want2 = oracle.assemble_source_vector(oasm, sdim, values=vals)
            got2 = fa.VectorAssembler().assemble_vector(asm2)
            assert engine.last_kernel_name() == "k_source_elements_tiled + k_vector_from_partials", name
            assert np.abs(got2 - want2).max() <= 1e-12 * np.abs(want2).max(), name
    finally:
        engine.close()


def test_tiles_survive_a_vertex_update(oracle):
    """the tables depend on the connectivity only: fh_update_vertices keeps them (no rebuild), and the residual follows the new coordinates"""
    rng = np.random.default_rng(21)
    m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 7, 6, 5, 1)
    w, p = quadrature.tensor.hexahedron_gauss(2)
    u = 0.004 * rng.standard_normal(3 * m.num_nodes())
    engine = fa.Engine(0)
    try:
        asm = _assembler(engine, m, "NEO_HOOKEAN", w, p, u)
        first = fa.VectorAssembler().assemble_vector(asm)
        st, want = _oracle_vector(oracle, oracle.HEX8, "NEO_HOOKEAN", m, m.connectivity, w, p, u)
        assert st == 0 and np.abs(first - want).max() <= 1e-12 * np.abs(want).max()
        moved = m.vertices + 0.03 * rng.uniform(-1, 1, m.vertices.shape)
        engine.update_vertices(moved)
        got = fa.VectorAssembler().assemble_vector(asm)
        assert engine.last_kernel_name() == TILED
        m2 = fa.Mesh(moved, m.connectivity, fa.HEX8)
        st, want2 = _oracle_vector(oracle, oracle.HEX8, "NEO_HOOKEAN", m2, m2.connectivity, w, p, u)
        assert st == 0 and np.abs(got - want2).max() <= 1e-12 * np.abs(want2).max()
        assert np.abs(want2 - want).max() > 1e-3 * np.abs(want).max()      # (the update did change the answer)
    finally:
        engine.close()


@pytest.mark.parametrize("opname", ["LAPLACE", "LINEAR_ELASTIC"])
def test_quadrature_free_residual_of_affine_hex8_meshes(oracle, opname):
    """Round 5: when every Hex8 element is a parallelepiped, the operator is linear (Laplace, LinearElastic with one parameter pair) and the rule is
    symmetric in every coordinate, the residual's element pass integrates the seven monomial terms of the integrand with the rule's moments
    instead of looping over the points (element_pass.hpp, AFFM = 2).  Against the oracle (elliptic.rs:457-531 restated) at 1e-12 and against the
    point loop of the same library (FENRIS_HIP_NO_MOMENT_RESIDUAL=1) on sheared, graded and mirrored boxes, for the 8- and the 27-point Gauss
    rule; a rule with one point moved (odd moments no longer vanish) and parameters that differ from point to point keep the point loop and
    must give the oracle's vector as well."""
    rng = np.random.default_rng(23)
    box = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 7, 6, 8, 1)
    A = np.array([[1.0, 0.3, 0.1], [0.0, 0.8, -0.2], [0.25, 0.0, 1.4]])
    meshes = {
        "box": box,
        "sheared": fa.Mesh(box.vertices @ A.T + np.array([3.0, -1.0, 0.5]), box.connectivity, fa.HEX8),
        "graded": fa.Mesh(np.stack([box.vertices[:, 0] ** 1.7, box.vertices[:, 1] ** 0.8 * 2.0, np.expm1(box.vertices[:, 2])], axis=1),
                          box.connectivity, fa.HEX8),
        "mirrored": fa.Mesh(box.vertices * np.array([-1.0, 1.0, 1.0]), box.connectivity, fa.HEX8),
    }
    s = 1 if opname == "LAPLACE" else 3
    rules = {"gauss2": quadrature.tensor.hexahedron_gauss(2), "gauss3": quadrature.tensor.hexahedron_gauss(3)}
    w2, p2 = rules["gauss2"]
    p_moved = np.array(p2, dtype=np.float64).copy()
    p_moved[3, 0] += 0.05
    rules["gauss2, one point moved"] = (w2, p_moved)
    for mname, mesh in meshes.items():
        u = rng.uniform(-1, 1, s * mesh.num_nodes())
        for rname, (w, p) in rules.items():
            got, energy = {}, {}
            for no_moments in (0, 1):
                eng = fa.Engine(0)
                try:
                    eng.set_option("FENRIS_HIP_NO_MOMENT_RESIDUAL", no_moments)
                    asm = _assembler(eng, mesh, opname, w, p, u)
                    got[no_moments] = np.asarray(fa.VectorAssembler().assemble_vector(asm)).copy()
                    assert eng.last_kernel_name() == TILED
                    energy[no_moments] = fa.assemble_scalar(asm)     # the energy takes the same route (a quadratic form: its parts do not mix)
                finally:
                    eng.close()
            st, want = _oracle_vector(oracle, oracle.HEX8, opname, mesh, np.asarray(mesh.connectivity), w, p, u)
            assert st == 0
            scale = np.abs(want).max()
            assert np.abs(got[0] - want).max() <= 1e-12 * scale, (mname, rname)
            assert np.abs(got[1] - want).max() <= 1e-12 * scale, (mname, rname)
            assert np.abs(got[0] - got[1]).max() <= 1e-13 * scale, (mname, rname)
            ref = oracle.ElementAssembler(oracle.HEX8, getattr(oracle, opname), mesh.vertices, np.asarray(mesh.connectivity), w, p,
                                          params=(LAME.as_pair() if opname != "LAPLACE" else None), u=u)
            st, _, eo = oracle.assemble_scalar(ref)
            assert st == 0
            assert abs(energy[0] - eo) <= 1e-12 * abs(eo) and abs(energy[1] - eo) <= 1e-12 * abs(eo), (mname, rname, energy, eo)
    if opname == "LINEAR_ELASTIC":   # parameters that differ from point to point: the point loop
        mesh = meshes["sheared"]
        u = rng.uniform(-1, 1, 3 * mesh.num_nodes())
        pairs = np.stack([3.0e2 * (1.0 + 0.1 * np.arange(len(w2))), 5.0e2 * (1.0 - 0.05 * np.arange(len(w2)))], axis=1)
        eng = fa.Engine(0)
        try:
            qt = fa.UniformQuadratureTable.from_points_and_weights(p2, w2).with_data([fa.LameParameters(a, b) for a, b in pairs])
            asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
                   .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(u).build())
            got = np.asarray(fa.VectorAssembler().assemble_vector(asm))
        finally:
            eng.close()
        ref = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, mesh.vertices, mesh.connectivity, w2, p2, params=pairs, u=u)
        st, _, want = oracle.assemble_vector(ref)
        assert st == 0
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
