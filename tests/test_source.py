"""ElementSourceAssembler (src/assembly/local/source.rs): oracle properties on the CPU, HIP parity on the GPU."""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

KIND = {"QUAD4": fa.QUAD4, "HEX8": fa.HEX8, "TET4": fa.TET4, "HEX27": fa.HEX27}


def _mesh(kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "QUAD4":
        m = fa.procedural.create_unit_square_uniform_quad_mesh_2d(5)
        h = 0.2
    elif kind == "HEX8":
        m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 4)
        h = 0.25
    elif kind == "HEX27":
        # distort the corner vertices only: Hex27's geometry is its embedded Hex8 (hexahedron.rs:324-326)
        m8 = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 2)
        m8 = fa.Mesh(m8.vertices + rng.uniform(-0.05, 0.05, m8.vertices.shape), m8.connectivity, m8.elem_kind)
        return fa.hex27_mesh_from_hex8(m8)
    else:
        m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
        h = 0.25
    return fa.Mesh(m.vertices + rng.uniform(-0.1 * h, 0.1 * h, m.vertices.shape), m.connectivity, m.elem_kind)


def _rule(kind):
    if kind == "QUAD4":
        return quadrature.tensor.quadrilateral_gauss(3)
    if kind == "HEX8":
        return quadrature.tensor.hexahedron_gauss(2)
    if kind == "HEX27":
        return quadrature.tensor.hexahedron_gauss(3)
    return quadrature.total_order.tetrahedron(2)


# ------------------------------------------------------------------------------------------- CPU: oracle pins
def test_oracle_source_vector_reproduces_inner_product(oracle):
    """tests/unit_tests/assembly/local/source.rs:19-111 on a Tet4 element (u linear, so the nodal interpolation is
    exact): u_K . f_K == int_K rho f . u, the right-hand side by a higher-order rule."""
    verts = np.array([[2.0, 0.0, 1.0], [3.0, 4.0, 1.0], [1.0, 1.0, 2.0], [3.0, 1.0, 4.0]])  # the reference's a, b, c, d
    conn = np.array([[0, 1, 2, 3]], dtype=np.uint64)

    def u(x):
        return np.stack([3 * x[..., 0] - 4 * x[..., 1] + 3 * x[..., 2] + 5, 3 * x[..., 0] - 2 * x[..., 1] + x[..., 2] - 3,
                         x[..., 0] + x[..., 1] - 2 * x[..., 2] + 1], axis=-1)

    def f(x):
        return np.stack([6 * x[..., 0] - 4 * x[..., 2] + 3, 2 * x[..., 0] + 3 * x[..., 1] - x[..., 2] + 5,
                         x[..., 1] - 0.5 * x[..., 2] + 2], axis=-1)

    rho = 1.7
    w2, p2 = oracle.tetrahedron_rule(2)
    asm = oracle.ElementAssembler(oracle.TET4, oracle.LAPLACE, verts, conn, w2, p2, params=[rho, 0.0])
    x = oracle.physical_quadrature_points(asm)
    st, fk = oracle.assemble_source_vector(asm, 3, values=rho * f(x))
    assert st == 0
    uk = u(verts).reshape(-1)
    # reference integral with the degree-3 rule (integrand has degree 2)
    w3, p3 = oracle.tetrahedron_rule(3)
    asm3 = oracle.ElementAssembler(oracle.TET4, oracle.LAPLACE, verts, conn, w3, p3)
    x3 = oracle.physical_quadrature_points(asm3)[0]
    vol = abs(np.linalg.det(verts[1:] - verts[0])) / 6.0
    detj = vol / (4.0 / 3.0)  # reference tet volume 4/3
    expected = float(np.sum(w3 * detj * rho * np.sum(f(x3) * u(x3), axis=-1)))
    assert abs(uk @ fk - expected) <= 1e-12 * max(1.0, abs(expected))


@pytest.mark.parametrize("kind", ["QUAD4", "HEX8", "TET4", "HEX27"])
def test_oracle_gravity_total_force_is_mass_times_g(oracle, kind):
    """sum of the nodal forces of rho g == rho g |Omega| (partition of unity)"""
    m = _mesh(kind)
    w, p = _rule(kind)
    d = m.vertices.shape[1]
    g = np.array([0.3, -9.81, 1.2])[:d]
    rho = 2.5
    asm = oracle.ElementAssembler(KIND[kind], oracle.LAPLACE, m.vertices, m.connectivity, w, p, params=[rho, 0.0])
    st, out = oracle.assemble_source_vector(asm, d, g=g)
    assert st == 0
    total = out.reshape(-1, d).sum(axis=0)
    # boundary vertices move with the distortion, so measure |Omega| with the same rule: sum_q w |det J|
    st, vol = oracle.assemble_source_vector(asm, 1, g=np.array([1.0 / rho]))
    assert st == 0
    np.testing.assert_allclose(total, rho * g * vol.sum(), rtol=1e-13)
    if kind != "HEX27":
        assert abs(vol.sum() - 1.0) < 0.2


def test_oracle_uniform_and_sampled_sources_agree(oracle):
    m = _mesh("HEX8")
    w, p = _rule("HEX8")
    g = np.array([0.0, 0.0, -9.81])
    rho = np.linspace(1.0, 2.0, len(w))
    params = np.stack([rho, np.zeros_like(rho)], axis=1)
    asm = oracle.ElementAssembler(oracle.HEX8, oracle.LAPLACE, m.vertices, m.connectivity, w, p, params=params)
    st, a = oracle.assemble_source_vector(asm, 3, g=g)
    vals = np.broadcast_to(rho[None, :, None] * g[None, None, :], (asm.E, len(w), 3))
    st2, b = oracle.assemble_source_vector(asm, 3, values=vals)
    assert st == 0 and st2 == 0
    np.testing.assert_allclose(a, b, rtol=1e-14, atol=1e-14)


# ------------------------------------------------------------------------------------------- GPU parity
@pytest.fixture(scope="module")
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["QUAD4", "HEX8", "TET4", "HEX27"])
def test_physical_quadrature_points_match_oracle(engine, oracle, kind):
    m = _mesh(kind)
    w, p = _rule(kind)
    engine.set_mesh(m)
    engine.set_quadrature_uniform(w, p, None)
    x = engine.physical_quadrature_points(len(w))
    asm = oracle.ElementAssembler(KIND[kind], oracle.LAPLACE, m.vertices, m.connectivity, w, p)
    np.testing.assert_allclose(x, oracle.physical_quadrature_points(asm), rtol=0, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["QUAD4", "HEX8", "TET4", "HEX27"])
@pytest.mark.parametrize("sdim", ["scalar", "vector"])
def test_gravity_source_matches_oracle(engine, oracle, kind, sdim):
    m = _mesh(kind, seed=3)
    w, p = _rule(kind)
    d = m.vertices.shape[1]
    s = 1 if sdim == "scalar" else d
    g = np.array([0.3, -9.81, 1.2])[:s]
    rho = np.linspace(1.0, 2.0, len(w))
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_data([fa.Density(r) for r in rho])
    asm = (fa.ElementSourceAssemblerBuilder.new(engine).with_finite_element_space(m)
           .with_source(fa.GravitySource.from_acceleration(g)).with_quadrature_table(qt).build())
    out = fa.VectorAssembler().assemble_vector(asm)
    oasm = oracle.ElementAssembler(KIND[kind], oracle.LAPLACE, m.vertices, m.connectivity, w, p,
                                   params=np.stack([rho, np.zeros_like(rho)], axis=1))
    st, ref = oracle.assemble_source_vector(oasm, s, g=g)
    assert st == 0
    assert np.max(np.abs(out - ref)) <= 1e-12 * np.max(np.abs(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["QUAD4", "HEX8", "TET4", "HEX27"])
def test_sampled_source_matches_oracle_and_accumulates(engine, oracle, kind):
    m = _mesh(kind, seed=5)
    w, p = _rule(kind)
    d = m.vertices.shape[1]

    def f(x, _data):
        return np.stack([np.sin(x[..., 0]) + x[..., 1] ** 2, np.cos(x[..., 1] * x[..., 0])] +
                        ([x[..., 2] * x[..., 0] - 1.0] if d == 3 else []), axis=-1)

    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    asm = (fa.ElementSourceAssemblerBuilder.new(engine).with_finite_element_space(m)
           .with_source(fa.SourceFunction(d, f)).with_quadrature_table(qt).build())
    out = np.full(d * m.num_nodes(), 0.5)  # assemble_vector_into accumulates (global.rs:582-608)
    fa.VectorAssembler().assemble_vector_into(out, asm)
    oasm = oracle.ElementAssembler(KIND[kind], oracle.LAPLACE, m.vertices, m.connectivity, w, p)
    st, ref = oracle.assemble_source_vector(oasm, d, values=f(oracle.physical_quadrature_points(oasm), None),
                                            out=np.full(d * m.num_nodes(), 0.5))
    assert st == 0
    assert np.max(np.abs(out - ref)) <= 1e-12 * np.max(np.abs(ref))


@pytest.mark.gpu
def test_source_vector_device_resident(engine, oracle):
    import torch

    m = _mesh("HEX8", seed=7)
    w, p = _rule("HEX8")
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.Density(3.0))
    asm = (fa.ElementSourceAssemblerBuilder.new(engine).with_finite_element_space(m)
           .with_source(fa.GravitySource([0.0, 0.0, -9.81])).with_quadrature_table(qt).build())
    out = torch.zeros(3 * m.num_nodes(), dtype=torch.float64, device="cuda:0")
    fa.VectorAssembler().assemble_vector_into(out, asm)
    oasm = oracle.ElementAssembler(oracle.HEX8, oracle.LAPLACE, m.vertices, m.connectivity, w, p, params=[3.0, 0.0])
    st, ref = oracle.assemble_source_vector(oasm, 3, g=[0.0, 0.0, -9.81])
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= 1e-12 * np.max(np.abs(ref))


@pytest.mark.gpu
def test_source_vector_argument_errors(engine):
    m = _mesh("HEX8")
    w, p = _rule("HEX8")
    engine.set_mesh(m)
    engine.set_quadrature_uniform(w, p, None)
    out = np.zeros(3 * m.num_nodes())
    with pytest.raises(fa.FenrisError):
        engine.assemble_source_vector(out, 3, g=[0.0, 0.0, 1.0])  # uniform source without a density table
    with pytest.raises(fa.FenrisError):
        engine.assemble_source_vector(np.zeros(2 * m.num_nodes()), 2, values=np.zeros((m.num_elements(), len(w), 2)))
