"""Patch test: a value pin that does not go through the oracle (and one for the oracle itself).

For a linear displacement field u(x) = A x + c the strain is constant, the stress is sigma = lambda tr(eps) I + 2 mu eps with
eps = sym(A) (LinearElasticMaterial, fenris-solid/src/materials.rs:83-123), and K u is the vector of nodal forces of that stress:
  * zero at every node inside the domain (a constant stress is in equilibrium),
  * sum_i f_i (x) x_i = sigma |Omega|  (virtual work with the linear test fields w = B x, which lie in every Lagrange space),
  * on a box whose boundary nodes form a uniform grid: f_i = sum over the faces through node i of sigma n x (tributary area).
Every entry of K takes part, both Lame parameters and the scale of the quadrature weights are pinned, and nothing is compared with
another implementation.  The same for the Laplace operator with u = a . x (laplace.rs:26-73): sum_i f_i x_i = a |Omega|.
Element kinds: Hex8 on a uniform box (the affine-element kernel) and with the interior vertices moved (the general kernel), Tet4,
Hex27, Tet10.
"""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

E_MOD, NU = 1e6, 0.2
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(E_MOD, NU))
A = np.array([[0.011, -0.004, 0.007], [0.003, -0.009, 0.002], [-0.006, 0.005, 0.013]])
C0 = np.array([0.3, -0.2, 0.1])
GRAD = np.array([0.7, -1.3, 0.4])


def _sigma():
    mu, lam = LAME.as_pair()
    eps = 0.5 * (A + A.T)
    return lam * np.trace(eps) * np.eye(3) + 2.0 * mu * eps


def _cases():
    cells = 6
    hex8 = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, cells)
    rng = np.random.default_rng(11)
    v = hex8.vertices.copy()
    inside = np.all((v > 1e-9) & (v < 1 - 1e-9), axis=1)
    v[inside] += (0.2 / cells) * rng.uniform(-1, 1, (int(inside.sum()), 3))
    tet4 = fa.procedural.create_unit_box_uniform_tet_mesh_3d(3)
    return {
        "hex8_uniform": (hex8, quadrature.tensor.hexahedron_gauss(2), True),
        "hex8_moved_interior": (fa.Mesh(v, hex8.connectivity, fa.HEX8), quadrature.tensor.hexahedron_gauss(2), True),
        "tet4": (tet4, quadrature.total_order.tetrahedron(1), False),
        "hex27": (fa.hex27_mesh_from_hex8(fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 3)), quadrature.tensor.hexahedron_gauss(3), False),
        "tet10": (fa.tet10_mesh_from_tet4(fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)), quadrature.total_order.tetrahedron(2), False),
    }, cells


def _check_elasticity(mesh, K, uniform_boundary, cells):
    x = mesh.vertices
    u = (x @ A.T + C0).reshape(-1)
    f = (K @ u).reshape(-1, 3)
    sig = _sigma()
    scale = np.abs(f).max()
    inside = np.all((x > 1e-9) & (x < 1 - 1e-9), axis=1)
    assert inside.any() and np.abs(f[inside]).max() <= 2e-11 * scale            # equilibrium inside
    moment = f.T @ x                                                             # sum_i f_i x_i^T = sigma |Omega|
    assert np.abs(moment - sig).max() <= 1e-11 * np.abs(sig).max()
    assert np.abs(f.sum(axis=0)).max() <= 1e-11 * scale                          # no net force
    if uniform_boundary:                                                         # nodal forces of the boundary traction, node by node
        h = 1.0 / cells
        idx = np.rint(x / h).astype(int)
        wgt = np.where((idx == 0) | (idx == cells), 0.5 * h, h)                  # tributary length per direction
        ref = np.zeros_like(f)
        for ax in range(3):
            o = [a for a in range(3) if a != ax]
            area = wgt[:, o[0]] * wgt[:, o[1]]
            for side, sign in ((0, -1.0), (cells, 1.0)):
                on = idx[:, ax] == side
                ref[on] += np.outer(area[on], sign * sig[:, ax])
        on_boundary = ~inside
        assert np.abs(f[on_boundary] - ref[on_boundary]).max() <= 2e-11 * scale


def _check_laplace(mesh, K):
    x = mesh.vertices
    f = K @ (x @ GRAD + 0.25)
    scale = np.abs(f).max()
    inside = np.all((x > 1e-9) & (x < 1 - 1e-9), axis=1)
    assert np.abs(f[inside]).max() <= 2e-11 * scale
    assert np.abs(f @ x - GRAD).max() <= 1e-11 * np.abs(GRAD).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hex8_uniform", "hex8_moved_interior", "tet4", "hex27", "tet10"])
def test_patch_linear_elasticity_on_the_device(name):
    cases, cells = _cases()
    mesh, (w, p), uniform = cases[name]
    asm = (fa.ElementEllipticAssemblerBuilder().with_finite_element_space(mesh)
           .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
           .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)).with_u(None).build())
    K = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm).to_scipy()
    expected = {"hex8_uniform": "k_affine_rows", "hex8_moved_interior": "k_hex8_rows"}.get(name)
    if expected:
        assert expected in asm.engine.last_kernel_name()   # the two Hex8 kernels of the benchmark are the ones pinned here
    _check_elasticity(mesh, K, uniform, cells)
    # the coloured scatter (CsrParAssembler semantics) gives the same forces
    Kc = fa.CsrParAssembler().assemble(fa.color_nodes(asm), asm).to_scipy() if hasattr(fa, "CsrParAssembler") else None
    if Kc is not None:
        _check_elasticity(mesh, Kc, uniform, cells)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hex8_uniform", "hex8_moved_interior", "tet4", "hex27"])
def test_patch_laplace_on_the_device(name):
    cases, _ = _cases()
    mesh, (w, p), _ = cases[name]
    asm = (fa.ElementEllipticAssemblerBuilder().with_finite_element_space(mesh).with_operator(fa.LaplaceOperator())
           .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w)).with_u(None).build())
    _check_laplace(mesh, fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm).to_scipy())


@pytest.mark.parametrize("name", ["hex8_uniform", "hex8_moved_interior", "tet4", "hex27"])
def test_patch_linear_elasticity_oracle(oracle, name):
    """the same pin for the CPU restatement (test infrastructure): its values are anchored without the device"""
    import scipy.sparse as sp

    cases, cells = _cases()
    mesh, (w, p), uniform = cases[name]
    kind = {"hex8_uniform": oracle.HEX8, "hex8_moved_interior": oracle.HEX8, "tet4": oracle.TET4, "hex27": oracle.HEX27}[name]
    ref = oracle.ElementAssembler(kind, oracle.LINEAR_ELASTIC, mesh.vertices, mesh.connectivity, w, p, params=LAME.as_pair())
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    n = 3 * mesh.num_nodes()
    K = sp.csr_matrix((vals, ci.astype(np.int64), ro.astype(np.int64)), shape=(n, n))
    _check_elasticity(mesh, K, uniform, cells)
