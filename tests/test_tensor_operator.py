"""FH_TENSOR: an elliptic operator given as data -- C(a, b)[i][k] = sum_jl a[j] A[i][j][k][l] b[l], one tensor per quadrature point --, symmetric
or not (src/assembly/operators.rs:146-189: `contract`, `symmetry`, the fill rule of `accumulate_contractions_into` :176-181; elliptic.rs:434-436:
clone_upper_to_lower only for Symmetry::Symmetric).  The reference exercises the NonSymmetric branch with a closure-defined mock operator
(tests/unit_tests/assembly/local/elliptic.rs:142-185, 331-392: K_e is the Jacobian of the element vector); here the operator is data and linear,
so its element vector is f(u) = K_e u and the same statement is checked against a direct evaluation."""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))


def _tet10_of_the_reference_test(oracle):
    # tests/unit_tests/assembly/local/elliptic.rs:336-343
    v4 = np.array([[2.0, 0.0, 1.0], [3.0, 4.0, 1.0], [1.0, 1.0, 2.0], [3.0, 1.0, 4.0]])
    return oracle.refine_to_quadratic(oracle.TET4, v4, np.array([[0, 1, 2, 3]], dtype=np.uint64))


def _direct_element_matrix(oracle, kind, geom_kind, ng, verts, conn_e, w, p, A):
    """K[(I, i), (J, k)] = sum_q w |det J| sum_jl g_I[j] A_q[i, j, k, l] g_J[l], written out with numpy"""
    ev = verts[conn_e.astype(int)]
    n, d = len(conn_e), verts.shape[1]
    K = np.zeros((d * n, d * n))
    for q in range(len(w)):
        G = np.asarray(oracle.element_gradients(kind, p[q])).T        # (n, d) reference gradients
        Gg = np.asarray(oracle.element_gradients(geom_kind, p[q])).T  # (ng, d)
        J = ev[:ng].T @ Gg
        g = G @ np.linalg.inv(J)
        K += w[q] * abs(np.linalg.det(J)) * np.einsum("Ij,ijkl,Jl->IiJk", g, A[q], g).reshape(d * n, d * n)
    return K


def test_oracle_nonsymmetric_fill_on_the_reference_tet10(oracle):
    v, c = _tet10_of_the_reference_test(oracle)
    w, p = oracle.tetrahedron_rule(4)
    A = np.random.default_rng(0).standard_normal((len(w), 3, 3, 3, 3))
    want = _direct_element_matrix(oracle, oracle.TET10, oracle.TET4, 4, v, c[0], w, p, A)
    asm = oracle.ElementAssembler(oracle.TET10, oracle.TENSOR, v, c, w, p, tensor=A, tensor_symmetric=False)
    st, ke = asm.element_matrix(0)
    assert st == 0 and np.abs(ke - want).max() <= 1e-13 * np.abs(want).max()
    assert np.abs(ke - ke.T).max() > 0.1 * np.abs(ke).max()     # every block of every row is there, and they are not mirror images
    # K_e is the Jacobian of the element vector f(u) = K_e u of this LINEAR operator: central differences of the direct evaluation
    u0 = np.random.default_rng(1).standard_normal(30)
    h = 1e-6
    jac = np.stack([(want @ (u0 + h * e) - want @ (u0 - h * e)) / (2 * h) for e in np.eye(30)], axis=1)
    assert np.abs(jac - ke).max() <= 1e-6 * np.abs(ke).max()
    # Symmetry::Symmetric with the SAME data: only I <= J is formed and mirrored (operators.rs:176-179, util.rs:38-51)
    asm_s = oracle.ElementAssembler(oracle.TET10, oracle.TENSOR, v, c, w, p, tensor=A, tensor_symmetric=True)
    st, ks = asm_s.element_matrix(0)
    assert st == 0 and np.array_equal(ks, ks.T) and np.abs(np.triu(ks) - np.triu(want)).max() <= 1e-13 * np.abs(want).max()


def test_oracle_tensor_of_linear_elasticity_is_the_material(oracle):
    """the index convention, pinned by an operator the reference ships: A = mu (d_ik d_jl + d_il d_jk) + lambda d_ij d_kl is LinearElasticMaterial
    (fenris-solid/src/materials.rs:108-118), whose oracle restatement is pinned by the reference's own vectors (tests/test_oracle_kat.py)"""
    vh, ch = oracle.unit_box_hex_mesh(2)
    rng = np.random.default_rng(3)
    vh = vh + 0.05 * rng.uniform(-1, 1, vh.shape)
    w, p = oracle.hexahedron_gauss(2)
    A = fa.TensorEllipticOperator.linear_elastic(LAME.mu, LAME.lambda_)
    le = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, vh, ch, w, p, params=np.array(LAME.as_pair()))
    for sym in (True, False):
        te = oracle.ElementAssembler(oracle.HEX8, oracle.TENSOR, vh, ch, w, p, tensor=A, tensor_symmetric=sym)
        for e in (0, 5):
            k1, k2 = le.element_matrix(e)[1], te.element_matrix(e)[1]
            assert np.abs(k1 - k2).max() <= 1e-13 * np.abs(k1).max()


# ------------------------------------------------------------------------------------------------------------------ device
def _hex8_mesh():
    m = fa.procedural.create_unit_box_uniform_hex_mesh_3d(3)
    rng = np.random.default_rng(5)
    return fa.Mesh(m.vertices + 0.04 * rng.uniform(-1, 1, m.vertices.shape), m.connectivity, m.elem_kind)


def _tet10_mesh():
    m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
    rng = np.random.default_rng(6)
    m = fa.Mesh(m.vertices + 0.03 * rng.uniform(-1, 1, m.vertices.shape), m.connectivity, m.elem_kind)
    return fa.tet10_mesh_from_tet4(m)


CASES = {"hex8": (_hex8_mesh, lambda: quadrature.tensor.hexahedron_gauss(2), "HEX8"),
         "tet10": (_tet10_mesh, lambda: quadrature.total_order.tetrahedron(4), "TET10")}


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["hex8", "tet10"])
@pytest.mark.parametrize("symmetric", [False, True])
def test_device_tensor_operator_matches_the_oracle(oracle, case, symmetric):
    mesh_fn, rule_fn, okind = CASES[case]
    mesh = mesh_fn()
    w, p = rule_fn()
    rng = np.random.default_rng(11)
    A = rng.standard_normal((len(w), 3, 3, 3, 3))
    if symmetric:
        A = A + A.transpose(0, 3, 4, 1, 2)       # A[i][j][k][l] == A[k][l][i][j]
    ref = oracle.ElementAssembler(getattr(oracle, okind), oracle.TENSOR, mesh.vertices, mesh.connectivity, w, p, tensor=A,
                                  tensor_symmetric=symmetric)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    eng = fa.Engine(0)
    try:
        qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
        asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
               .with_operator(fa.TensorEllipticOperator(A, symmetric=symmetric)).with_quadrature_table(qt).with_u(None).build())
        assert asm.solution_dim() == 3
        seen = set()
        for mode in (fa.SCATTER_GATHER, fa.SCATTER_ATOMIC, fa.SCATTER_COLORED):
            k = (fa.CsrParAssembler().assemble(fa.color_nodes(asm), asm) if mode == fa.SCATTER_COLORED else fa.CsrAssembler(mode).assemble(asm))
            seen.add(eng.last_kernel_name())
            assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
            assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max(), (mode, eng.last_kernel_name())
            a = k.to_scipy()
            d = abs(a - a.T).max()
            if symmetric:
                assert d <= 1e-12 * np.abs(vals).max()
            else:
                assert d > 1e-3 * np.abs(vals).max()      # full-row fill: the lower blocks are not mirror images
        # the element matrices (ElementMatrixAssembler::assemble_element_matrix_into) as well
        ke = eng.element_matrices(0, mesh.num_elements())
        for e in (0, mesh.num_elements() - 1):
            want = ref.element_matrix(e)[1]
            assert np.abs(ke[e] - want).max() <= 1e-12 * np.abs(want).max()      # [e][row][col]
        if case == "tet10":
            assert any(("k_rows_from_dense" if not symmetric else "k_rows_from_tri") in s for s in seen), seen     # ten nodes: the two-pass form (K_e in full and transposed for the non-symmetric tensor, its triangle for the symmetric one)
    finally:
        eng.close()


@pytest.mark.gpu
def test_device_tensor_of_linear_elasticity_is_the_material(oracle):
    mesh = _hex8_mesh()
    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    eng = fa.Engine(0)
    try:
        a_le = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
                .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt.with_uniform_data(LAME)).with_u(None).build())
        k_le = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(a_le)
        a_t = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
               .with_operator(fa.TensorEllipticOperator(fa.TensorEllipticOperator.linear_elastic(LAME.mu, LAME.lambda_), symmetric=True))
               .with_quadrature_table(qt).with_u(None).build())
        k_t = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(a_t)
        assert np.abs(k_t.values - k_le.values).max() <= 1e-12 * np.abs(k_le.values).max()
        # residual and energy of a data-defined operator are not behind the ABI (matrix only): a clear answer, not a wrong vector
        with pytest.raises(fa.FenrisError):
            fa.VectorAssembler().assemble_vector(a_t)
    finally:
        eng.close()
