"""Conjugate gradients (fenris-sparse/src/cg.rs) and error estimation (src/error.rs): oracle pins on the CPU, HIP
parity and the closed MMS loop (assembly -> Dirichlet -> CG -> error norms, all on the device) on the GPU."""
import json
import os

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature
from conftest import GOLDEN

KIND = {"QUAD4": fa.QUAD4, "HEX8": fa.HEX8, "TET4": fa.TET4, "HEX27": fa.HEX27}


def _spd_csr(n, seed=0):
    rng = np.random.default_rng(seed)
    import scipy.sparse as sp

    a = sp.random(n, n, density=0.08, random_state=seed, format="csr")
    a = a + a.T + sp.diags(np.full(n, 4.0) + rng.uniform(0, 1, n))
    a = a.tocsr()
    a.sort_indices()
    return a


# ------------------------------------------------------------------------------------------- CPU: oracle pins
@pytest.mark.parametrize("jacobi", [False, True])
def test_oracle_cg_solves_spd_system(oracle, jacobi):
    a = _spd_csr(60)
    x_true = np.linspace(-1, 1, 60)
    b = a @ x_true
    st, x, it = oracle.cg_solve(a.indptr, a.indices, a.data, b, jacobi=jacobi, tol=1e-12, max_iter=1000)
    assert st == 0 and 0 < it <= 60 + 5
    np.testing.assert_allclose(x, x_true, atol=1e-9)
    # CG's own residual criterion (cg.rs:108-124)
    assert np.linalg.norm(b - a @ x) <= 1e-10 * np.linalg.norm(b)


def test_oracle_cg_edge_cases(oracle):
    a = _spd_csr(30)
    b = np.ones(30)
    # zero right-hand side: x is overwritten with zeros, no iterations (cg.rs:409-412)
    st, x, it = oracle.cg_solve(a.indptr, a.indices, a.data, np.zeros(30), x0=np.ones(30))
    assert st == 0 and it == 0 and np.all(x == 0.0)
    # exact initial guess: converged before the first update
    st, x, it = oracle.cg_solve(a.indptr, a.indices, a.data, a @ b, x0=b)
    assert st == 0 and it == 0
    # max_iter reached -> MaxIterationsReached (code 7), iterate so far is returned
    st, x, it = oracle.cg_solve(a.indptr, a.indices, a.data, b, tol=1e-14, max_iter=2)
    assert st == 7 and it == 2
    # negative definite operator -> IndefiniteOperator (code 8) at the first iteration
    st, x, it = oracle.cg_solve(a.indptr, a.indices, -a.data, b, jacobi=False)
    assert st == 8 and it == 0
    # negative diagonal preconditioner with an SPD operator -> z.r <= 0 -> IndefinitePreconditioner (code 9)
    import scipy.sparse as sp

    m = sp.csr_matrix(np.array([[2.0, -3.0], [-3.0, -1.0]]))  # diagonal (2, -1), indefinite anyway: p.Ap decides first
    st, _, _ = oracle.cg_solve(m.indptr, m.indices, m.data, np.array([1.0, 1.0]), jacobi=True)
    assert st in (8, 9)


@pytest.mark.parametrize("kind", ["QUAD4", "HEX8", "TET4", "HEX27"])
def test_oracle_error_norms_of_interpolated_polynomials(oracle, kind):
    """u_h = nodal interpolant of a function inside the element space => both errors vanish; shifting u_h by a
    constant c gives L2^2 = c^2 |Omega| and leaves the H1 seminorm at zero."""
    if kind == "QUAD4":
        m, (w, p) = fa.procedural.create_unit_square_uniform_quad_mesh_2d(3), quadrature.tensor.quadrilateral_gauss(3)
    elif kind == "HEX8":
        m, (w, p) = fa.procedural.create_unit_box_uniform_hex_mesh_3d(2), quadrature.tensor.hexahedron_gauss(3)
    elif kind == "HEX27":
        m = fa.hex27_mesh_from_hex8(fa.procedural.create_unit_box_uniform_hex_mesh_3d(2))
        w, p = quadrature.tensor.hexahedron_gauss(3)
    else:
        m, (w, p) = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2), quadrature.total_order.tetrahedron(2)
    d = m.vertices.shape[1]
    coef = np.array([0.7, -1.3, 2.1])[:d]

    def u(x):
        return (x @ coef + 0.4)[..., None]

    asm = oracle.ElementAssembler(KIND[kind], oracle.LAPLACE, m.vertices, m.connectivity, w, p)
    xq = oracle.physical_quadrature_points(asm)
    uh = u(m.vertices)[:, 0]
    grad = np.broadcast_to(coef[None, None, :, None], xq.shape[:2] + (d, 1))
    st, l2 = oracle.estimate_error_squared(asm, 0, 1, uh, u(xq))
    st2, h1 = oracle.estimate_error_squared(asm, 1, 1, uh, grad)
    assert st == 0 and st2 == 0 and l2 < 1e-26 and h1 < 1e-24
    st, l2c = oracle.estimate_error_squared(asm, 0, 1, uh + 0.25, u(xq))
    st2, h1c = oracle.estimate_error_squared(asm, 1, 1, uh + 0.25, grad)
    assert abs(l2c - 0.0625) < 1e-13 and h1c < 1e-24


# ------------------------------------------------------------------------------------------- GPU parity
@pytest.fixture(scope="module")
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


def _elasticity_system(engine, cells=4, seed=0):
    """Hex8 linear elasticity with the x = 0 face clamped: SPD after the Dirichlet treatment"""
    import torch

    mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, cells)
    rng = np.random.default_rng(seed)
    mesh = fa.Mesh(mesh.vertices + rng.uniform(-0.02, 0.02, mesh.vertices.shape), mesh.connectivity, mesh.elem_kind)
    w, p = quadrature.tensor.hexahedron_gauss(2)
    lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e3, 0.3))
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh)
           .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt)
           .with_u(np.zeros(3 * mesh.num_nodes())).build())
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm, device_values=True)
    bc = np.where(mesh.vertices[:, 0] < 0.03)[0]
    fa.apply_homogeneous_dirichlet_bc_csr(k, bc, 3, asm)
    b = rng.standard_normal(3 * mesh.num_nodes())
    b.reshape(-1, 3)[bc] = 0.0
    return asm, k, b, torch


@pytest.mark.gpu
def test_spmv_matches_scipy(engine):
    asm, k, b, torch = _elasticity_system(engine)
    x = torch.from_numpy(b).cuda()
    y = torch.zeros_like(x)
    engine.spmv(k.values, x, y)
    ref = k.to_scipy() @ b
    assert np.max(np.abs(y.cpu().numpy() - ref)) <= 1e-13 * np.max(np.abs(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("pre", ["identity", "jacobi"])
def test_cg_matches_oracle(engine, oracle, pre):
    asm, k, b, torch = _elasticity_system(engine)
    a = k.to_scipy()
    st, x_ref, it_ref = oracle.cg_solve(a.indptr, a.indices, a.data, b, jacobi=(pre == "jacobi"), tol=1e-10, max_iter=5000)
    assert st == 0
    x = torch.zeros(len(b), dtype=torch.float64, device="cuda:0")
    cg = (fa.ConjugateGradient.new().with_operator(k, asm).with_max_iter(5000)
          .with_stopping_criterion(fa.RelativeResidualCriterion(1e-10)))
    if pre == "jacobi":
        cg = cg.with_preconditioner(fa.JacobiPreconditioner())
    it = cg.solve_with_guess(torch.from_numpy(b).cuda(), x)
    xs = x.cpu().numpy()
    assert abs(it - it_ref) <= max(3, it_ref // 20), (it, it_ref)  # summation order differs, the path does not
    assert np.linalg.norm(xs - x_ref) <= 1e-7 * np.linalg.norm(x_ref)
    assert np.linalg.norm(b - a @ xs) <= 2e-10 * np.linalg.norm(b)
    # bitwise reproducible: ordered reductions, no floating-point atomics
    x2 = torch.zeros_like(x)
    it2 = cg.solve_with_guess(torch.from_numpy(b).cuda(), x2)
    assert it2 == it and torch.equal(x, x2)


@pytest.mark.gpu
def test_cg_host_arrays_and_error_kinds(engine, oracle):
    asm, k, b, torch = _elasticity_system(engine, cells=3)
    vals = k.values.cpu().numpy()
    hk = fa.CsrMatrix(k.row_offsets, k.col_indices, vals)
    cg = (fa.ConjugateGradient.new().with_operator(hk, asm).with_preconditioner(fa.JacobiPreconditioner())
          .with_stopping_criterion(fa.RelativeResidualCriterion(1e-9)))
    x = np.zeros(len(b))
    it = cg.solve_with_guess(b, x)
    a = k.to_scipy()
    assert it > 0 and np.linalg.norm(b - a @ x) <= 2e-9 * np.linalg.norm(b)
    # zero right-hand side: x <- 0, no iterations
    x = np.ones(len(b))
    assert cg.solve_with_guess(np.zeros(len(b)), x) == 0 and np.all(x == 0.0)
    # MaxIterationsReached
    with pytest.raises(fa.CgSolveError) as ei:
        cg.with_max_iter(2).solve_with_guess(b, np.zeros(len(b)))
    assert ei.value.kind == "MaxIterationsReached" and ei.value.num_iterations == 2
    # IndefiniteOperator
    neg = fa.CsrMatrix(k.row_offsets, k.col_indices, -vals)
    with pytest.raises(fa.CgSolveError) as ei:
        (fa.ConjugateGradient.new().with_operator(neg, asm).with_stopping_criterion(fa.RelativeResidualCriterion(1e-9))
         .solve_with_guess(b, np.zeros(len(b))))
    assert ei.value.kind == "IndefiniteOperator"


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["QUAD4", "HEX8", "TET4", "HEX27"])
@pytest.mark.parametrize("sdim", ["scalar", "vector"])
def test_error_norms_match_oracle(engine, oracle, kind, sdim):
    rng = np.random.default_rng(11)
    if kind == "QUAD4":
        m, (w, p) = fa.procedural.create_unit_square_uniform_quad_mesh_2d(4), quadrature.tensor.quadrilateral_gauss(3)
    elif kind == "HEX8":
        m, (w, p) = fa.procedural.create_unit_box_uniform_hex_mesh_3d(3), quadrature.tensor.hexahedron_gauss(3)
    elif kind == "HEX27":
        m8 = fa.procedural.create_unit_box_uniform_hex_mesh_3d(2)
        m8 = fa.Mesh(m8.vertices + rng.uniform(-0.04, 0.04, m8.vertices.shape), m8.connectivity, m8.elem_kind)
        m = fa.hex27_mesh_from_hex8(m8)
        w, p = quadrature.tensor.hexahedron_gauss(3)
    else:
        m, (w, p) = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2), quadrature.total_order.tetrahedron(2)
    if kind != "HEX27":
        m = fa.Mesh(m.vertices + rng.uniform(-0.03, 0.03, m.vertices.shape), m.connectivity, m.elem_kind)
    d = m.vertices.shape[1]
    s = 1 if sdim == "scalar" else d

    def u(x):
        return np.stack([np.sin(2 * x[..., 0] + k) * np.cos(x[..., 1] - k) for k in range(s)], axis=-1)

    def gu(x):
        g = np.zeros(x.shape[:-1] + (d, s))
        for k in range(s):
            g[..., 0, k] = 2 * np.cos(2 * x[..., 0] + k) * np.cos(x[..., 1] - k)
            g[..., 1, k] = -np.sin(2 * x[..., 0] + k) * np.sin(x[..., 1] - k)
        return g

    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    asm = (fa.ElementSourceAssemblerBuilder.new(engine).with_finite_element_space(m)
           .with_source(fa.SourceFunction(s, lambda x, _d: u(x))).with_quadrature_table(qt).build())
    uh = (u(m.vertices) + 0.01 * rng.standard_normal((m.num_nodes(), s))).reshape(-1)
    l2 = fa.estimate_L2_error_squared(asm, u, uh)
    h1 = fa.estimate_H1_seminorm_error_squared(asm, gu, uh)
    oasm = oracle.ElementAssembler(KIND[kind], oracle.LAPLACE, m.vertices, m.connectivity, w, p)
    xq = oracle.physical_quadrature_points(oasm)
    st, l2o = oracle.estimate_error_squared(oasm, 0, s, uh, u(xq))
    st2, h1o = oracle.estimate_error_squared(oasm, 1, s, uh, gu(xq))
    assert st == 0 and st2 == 0
    assert abs(l2 - l2o) <= 1e-12 * l2o and abs(h1 - h1o) <= 1e-12 * h1o


@pytest.mark.gpu
def test_h1_error_reports_singular_jacobian(engine):
    m = fa.procedural.create_unit_box_uniform_hex_mesh_3d(2)
    v = m.vertices.copy()
    v[m.connectivity[3].astype(int)] = v[int(m.connectivity[3][0])]  # collapse one element
    m = fa.Mesh(v, m.connectivity, m.elem_kind)
    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    asm = (fa.ElementSourceAssemblerBuilder.new(engine).with_finite_element_space(m)
           .with_source(fa.SourceFunction(1, lambda x, _d: x[..., :1])).with_quadrature_table(qt).build())
    uh = np.zeros(m.num_nodes())
    assert fa.estimate_L2_error_squared(asm, lambda x: x[..., :1] * 0, uh) == 0.0  # only |det J| enters
    with pytest.raises(fa.SingularJacobianError):
        fa.estimate_H1_seminorm_error_squared(asm, lambda x: np.zeros(x.shape[:-1] + (3, 1)), uh)


# ------------------------------------------------------------------------------------------- MMS, closed on the device
@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,nres", [("poisson2d_mms_quad4_summary", "QUAD4", 5),
                                            ("poisson3d_mms_hex8_summary", "HEX8", 4),
                                            ("poisson3d_mms_tet4_summary", "TET4", 3),
                                            ("poisson3d_mms_hex27_summary", "HEX27", 3),
                                            ("poisson2d_mms_tri3_summary", "TRI3", 5)])
def test_mms_loop_on_device_matches_reference_errors(name, kind, nres):
    """tests/convergence_tests/poisson_{2d,3d}_mms.rs against reference_values/*.json (1 %,
    poisson_mms_common.rs:40-65): K, b, Dirichlet conditions, Jacobi-PCG (max_iter 10000, tol 1e-9 as in
    solve_linear_system :142-163) and both error norms all run on the device."""
    import torch

    ref = json.load(open(os.path.join(GOLDEN, "mms_reference_values.json")))["summaries"][name]
    if kind == "QUAD4":
        gen, rule, err_rule = (fa.procedural.create_unit_square_uniform_quad_mesh_2d, quadrature.tensor.quadrilateral_gauss(2),
                               quadrature.tensor.quadrilateral_gauss(6))
    elif kind == "HEX8":
        gen, rule, err_rule = (fa.procedural.create_unit_box_uniform_hex_mesh_3d, quadrature.tensor.hexahedron_gauss(2),
                               quadrature.tensor.hexahedron_gauss(6))
    elif kind == "HEX27":
        gen = lambda r: fa.hex27_mesh_from_hex8(fa.procedural.create_unit_box_uniform_hex_mesh_3d(r))
        rule, err_rule = quadrature.tensor.hexahedron_gauss(4), quadrature.tensor.hexahedron_gauss(6)
    elif kind == "TRI3":  # poisson_2d_mms.rs:99-105
        gen, rule, err_rule = (fa.procedural.create_unit_square_uniform_tri_mesh_2d, quadrature.total_order.triangle(0),
                               quadrature.total_order.triangle(6))
    else:
        t = json.load(open(os.path.join(GOLDEN, "tet_rule_6_24.json")))
        # poisson_3d_mms.rs:111-118: the reference assembles Tet4 with tetrahedron(0), errors with tetrahedron(6)
        gen, rule, err_rule = (fa.procedural.create_unit_box_uniform_tet_mesh_3d, quadrature.total_order.tetrahedron(0),
                               (np.array(t["weights"]), np.array(t["points"])))

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

    e_k, e_b, e_err = fa.Engine(0), fa.Engine(0), fa.Engine(0)
    try:
        for i, res in enumerate([1, 2, 4, 8, 16][:nres]):
            mesh = gen(res)
            w, p = rule
            d, N = mesh.vertices.shape[1], mesh.num_nodes()
            qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
            lap = (fa.ElementEllipticAssemblerBuilder(e_k).with_finite_element_space(mesh).with_operator(fa.LaplaceOperator())
                   .with_quadrature_table(qt).with_u(np.zeros(N)).build())
            k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(lap, device_values=True)
            src = (fa.ElementSourceAssemblerBuilder.new(e_b).with_finite_element_space(mesh)
                   .with_source(fa.SourceFunction(1, lambda x, _d: d * np.pi ** 2 * u_exact(x))).with_quadrature_table(qt).build())
            b = torch.zeros(N, dtype=torch.float64, device="cuda:0")
            fa.VectorAssembler().assemble_vector_into(b, src)
            bc = np.where(np.abs(mesh.vertices - 0.5).max(axis=1) > 0.4999)[0]
            fa.apply_homogeneous_dirichlet_bc_csr(k, bc, 1, lap)
            fa.apply_homogeneous_dirichlet_bc_rhs(b, bc, 1)
            u_h = torch.zeros(N, dtype=torch.float64, device="cuda:0")
            (fa.ConjugateGradient.new().with_operator(k, lap).with_preconditioner(fa.JacobiPreconditioner()).with_max_iter(10000)
             .with_stopping_criterion(fa.RelativeResidualCriterion(1e-9)).solve_with_guess(b, u_h))
            we, pe = err_rule
            err_asm = (fa.ElementSourceAssemblerBuilder.new(e_err).with_finite_element_space(mesh)
                       .with_source(fa.SourceFunction(1, lambda x, _d: u_exact(x)))
                       .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(pe, we)).build())
            uh = u_h.cpu().numpy()
            l2 = fa.estimate_L2_error(err_asm, u_exact, uh)
            h1 = fa.estimate_H1_seminorm_error(err_asm, u_grad, uh)
            assert abs(l2 - ref["L2_errors"][i]) / ref["L2_errors"][i] < 0.01, (res, l2, ref["L2_errors"][i])
            assert abs(h1 - ref["H1_seminorm_errors"][i]) / ref["H1_seminorm_errors"][i] < 0.01, (res, h1)
    finally:
        e_k.close(), e_b.close(), e_err.close()
