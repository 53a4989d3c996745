"""A second opinion on the oracle's VALUES: stiffness matrix, residual and energy restated once more, independently, in numpy -- written from
the formulas (K_e = sum_q w |det J| C(grad phi_I, grad phi_J), the contractions of laplace.rs:60-68,
materials.rs:108-118 / 287-315, the St. Venant-Kirchhoff tangent derived from P = F (2 mu E + lambda tr E I) by hand) with einsum over
whole meshes, dense matrices, its own shape functions and its own Gauss points, sharing no code with oracle/fenris_oracle.c.  The reference
holds no assembled global matrix to pin values against; this makes the pin two independent implementations agreeing to 1e-13 on
random distorted meshes, next to the chain energy KAT -> finite differences -> element matrices of test_oracle_kat.py."""
import numpy as np
import pytest

HEX_SIGNS = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=float)
QUAD_SIGNS = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=float)
# reference nodes of the triquadratic hexahedron: corners, edge midpoints, face centres, centre (hexahedron.rs:179-210: a table of data)
HEX27_NODES = np.array(list(map(tuple, HEX_SIGNS.astype(int))) + [
    (0, -1, -1), (-1, 0, -1), (-1, -1, 0), (1, 0, -1), (1, -1, 0), (0, 1, -1), (1, 1, 0), (-1, 1, 0), (0, -1, 1), (-1, 0, 1), (1, 0, 1), (0, 1, 1),
    (0, 0, -1), (0, -1, 0), (-1, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)], dtype=float)


def _quad1d(alpha, x):
    """value and derivative at x of the quadratic Lagrange polynomial of the node alpha in {-1, 0, 1}"""
    if alpha == 0:
        return 1.0 - x * x, -2.0 * x
    return 0.5 * x * (x + alpha), x + 0.5 * alpha
MU, LAM = 3.0e2, 5.0e2


def ref_gradients(kind, xi):
    """(n, d): gradients of the reference basis at xi; reference domains [-1, 1]^d, tetrahedron (-1,-1,-1), (1,-1,-1), (-1,1,-1), (-1,-1,1)"""
    if kind == "TET4":
        return np.array([[-0.5, -0.5, -0.5], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 0.5]])
    if kind == "HEX27":
        g = np.empty((27, 3))
        for n, node in enumerate(HEX27_NODES):
            v, dv = zip(*[_quad1d(node[k], xi[k]) for k in range(3)])
            g[n] = [dv[0] * v[1] * v[2], v[0] * dv[1] * v[2], v[0] * v[1] * dv[2]]
        return g
    sg = HEX_SIGNS if kind == "HEX8" else QUAD_SIGNS
    d = sg.shape[1]
    f = 1.0 + sg * np.asarray(xi)[None, :]                      # (n, d): the one-dimensional factors times 2
    g = np.empty_like(sg)
    for k in range(d):
        others = np.prod(np.delete(f, k, axis=1), axis=1)
        g[:, k] = sg[:, k] * others / 2.0 ** d
    return g


def rule(kind, oracle):
    if kind == "TET4":
        return oracle.tetrahedron_rule(2)
    x, w = np.polynomial.legendre.leggauss(3 if kind == "HEX27" else 2)
    d = 2 if kind == "QUAD4" else 3
    grids = np.meshgrid(*([x] * d), indexing="ij")
    pts = np.stack([g.reshape(-1) for g in grids], axis=1)
    wts = np.prod(np.stack([g.reshape(-1) for g in np.meshgrid(*([w] * d), indexing="ij")], axis=1), axis=1)
    return wts, pts


def contraction(op, a, b, F):
    """s x s block C(a, b) of one point; a, b: physical gradients of the row / column basis function"""
    d = len(a)
    I = np.eye(d)
    if op == "LAPLACE":
        return np.array([[a @ b]])
    if op == "LINEAR_ELASTIC":
        return MU * ((a @ b) * I + np.outer(b, a)) + LAM * np.outer(a, b)
    if op == "NEO_HOOKEAN":
        J = np.linalg.det(F)
        Fit = np.linalg.inv(F).T
        alpha = -MU + LAM * np.log(J)
        fa, fb = Fit @ a, Fit @ b
        return LAM * np.outer(fa, fb) - alpha * np.outer(fb, fa) + MU * (a @ b) * I
    # St. Venant-Kirchhoff: P = F S, S = 2 mu E + lambda tr(E) I, E = (F^T F - I) / 2;  dP = dF S + F dS
    E = 0.5 * (F.T @ F - I)
    S = 2 * MU * E + LAM * np.trace(E) * I
    return (b @ S @ a) * I + MU * (np.outer(F @ b, F @ a) + (a @ b) * (F @ F.T)) + LAM * np.outer(F @ a, F @ b)


def stress_and_energy(op, F, gu):
    d = gu.shape[0]
    I = np.eye(d)
    if op == "LAPLACE":
        return gu.T, 0.5 * float(np.sum(gu * gu))               # P (1 x d) = grad u^T
    if op == "LINEAR_ELASTIC":
        eps = 0.5 * (F + F.T) - I
        return 2 * MU * eps + LAM * np.trace(eps) * I, MU * np.sum(eps * eps) + 0.5 * LAM * np.trace(eps) ** 2
    if op == "NEO_HOOKEAN":
        J = np.linalg.det(F)
        C = F.T @ F
        return MU * (F - np.linalg.inv(F).T) + LAM * np.log(J) * np.linalg.inv(F).T, 0.5 * MU * (np.trace(C) - d) - MU * np.log(J) + 0.5 * LAM * np.log(J) ** 2
    E = 0.5 * (F.T @ F - I)
    return F @ (2 * MU * E + LAM * np.trace(E) * I), MU * np.sum(E * E) + 0.5 * LAM * np.trace(E) ** 2


def restate(kind, op, verts, conn, w, pts, u):
    """dense K, residual f and energy of the whole mesh"""
    N, d = verts.shape
    s = 1 if op == "LAPLACE" else d
    K = np.zeros((s * N, s * N))
    f = np.zeros(s * N)
    energy = 0.0
    for nodes in conn:
        X = verts[nodes]                                        # (n, d)
        U = u.reshape(-1, s)[nodes]                             # (n, s)
        for wq, xi in zip(w, pts):
            G = ref_gradients(kind, xi)                         # (n, d)
            if kind == "HEX27":                                 # sub-parametric: the geometry is the trilinear map of the eight corners
                Jm = X[:8].T @ ref_gradients("HEX8", xi)        # (hexahedron.rs:324-326)
            else:
                Jm = X.T @ G                                    # J[i][j] = sum_n x_n[i] d phi_n / d xi_j
            g = G @ np.linalg.inv(Jm)                           # (n, d): physical gradients  J^-T ghat
            scale = wq * abs(np.linalg.det(Jm))
            gu = g.T @ U                                        # (d, s): grad u = sum_n g_n u_n^T
            F = np.eye(d) + gu.T if s == d else None            # F = I + (grad u)^T
            P, psi = stress_and_energy(op, F, gu)
            energy += scale * psi
            for a_, I_ in enumerate(nodes):
                f[s * I_: s * I_ + s] += scale * (P @ g[a_])
                for b_, J_ in enumerate(nodes):
                    K[s * I_: s * I_ + s, s * J_: s * J_ + s] += scale * contraction(op, g[a_], g[b_], F)
    return K, f, energy


def small_mesh(kind, oracle, rng):
    if kind == "HEX27":
        v8, c8 = oracle.hex_mesh(1.0, 1, 1, 2, 1)
        v8 = np.asarray(v8, dtype=float).reshape(-1, 3)
        v8 = v8 + 0.08 * rng.uniform(-1, 1, v8.shape)              # (the corners carry the geometry; the other nodes follow)
        v, c = oracle.hex8_to_hex27(v8, c8)
        return np.asarray(v, dtype=float).reshape(-1, 3), np.asarray(c).astype(np.int64)
    if kind == "HEX8":
        v, c = oracle.hex_mesh(1.0, 2, 2, 1, 1)
        h = 1.0
    elif kind == "TET4":
        v, c = oracle.tet_mesh(1.0, 1, 1, 1, 1)
        h = 1.0
    else:
        v, c = oracle.quad_mesh(1.0, 1, 1, 3)
        h = 1.0 / 3
    v = np.asarray(v, dtype=float).reshape(-1, 2 if kind == "QUAD4" else 3)
    return v + 0.08 * h * rng.uniform(-1, 1, v.shape), np.asarray(c).astype(np.int64)


@pytest.mark.parametrize("kind,op", [(k, o) for k in ("HEX8", "TET4", "QUAD4") for o in ("LAPLACE", "LINEAR_ELASTIC", "NEO_HOOKEAN", "STVK")] +
                         [("HEX27", "LINEAR_ELASTIC"), ("HEX27", "NEO_HOOKEAN")])
def test_oracle_agrees_with_an_independent_numpy_restatement(oracle, kind, op):
    rng = np.random.default_rng(abs(hash((kind, op))) % 1000)
    verts, conn = small_mesh(kind, oracle, rng)
    d = verts.shape[1]
    s = 1 if op == "LAPLACE" else d
    w, pts = rule(kind, oracle)
    u = 0.05 * rng.standard_normal(s * len(verts)) if op in ("NEO_HOOKEAN", "STVK", "LAPLACE") else np.zeros(s * len(verts))
    K, f, energy = restate(kind, op, verts, conn, np.asarray(w), np.asarray(pts).reshape(len(w), d), u)
    asm = oracle.ElementAssembler(getattr(oracle, kind), getattr(oracle, op), verts, conn.astype(np.uint64), w, pts,
                                  params=None if op == "LAPLACE" else (MU, LAM), u=u)
    st, _, ro, ci, vals = oracle.assemble(asm)
    assert st == 0
    Ko = np.zeros_like(K)
    for r in range(len(ro) - 1):
        Ko[r, np.asarray(ci[ro[r]: ro[r + 1]]).astype(np.int64)] = vals[ro[r]: ro[r + 1]]
    assert np.abs(Ko - K).max() <= 1e-13 * np.abs(K).max(), (kind, op, np.abs(Ko - K).max() / np.abs(K).max())
    assert np.count_nonzero(K) <= len(vals)      # nothing outside the pattern (node pairs that share an element)
    st, _, fo = oracle.assemble_vector(asm)
    assert st == 0
    assert np.abs(fo - f).max() <= 1e-13 * max(np.abs(f).max(), 1e-300), (kind, op, "vector")
    res = oracle.assemble_scalar(asm)
    st, eo = res[0], res[-1]
    assert st == 0
    assert abs(eo - energy) <= 1e-13 * max(abs(energy), 1e-300), (kind, op, "energy")
