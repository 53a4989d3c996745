/*
 * fenris_oracle.c -- CPU restatement of the fenris assembly path.  TEST INFRASTRUCTURE ONLY
 * (see fenris_oracle.h for the rules and the parity-pin status).
 *
 * Written to follow the reference statement for statement -- including its redundant work (the
 * element is re-gathered and the shape gradients are re-evaluated for the Jacobian and again for
 * the basis gradients at every quadrature point, K_e is zero-filled per element, rows are scattered
 * with a forward linear search) -- because it doubles as the reported CPU baseline.
 * Compile with -ffp-contract=off.
 */
#include "fenris_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 27 /* max nodes / element */
#define MAXD 3

/* ------------------------------------------------------------------------------------------------
 * small dense helpers: nalgebra 0.32.1 semantics (un-vendored dependency; formulae restated)
 * ---------------------------------------------------------------------------------------------- */

/* column-major index */
#define CM(i, j, ld) ((size_t)(j) * (size_t)(ld) + (size_t)(i))

/* nalgebra Matrix::dot for dimension 2/3: a0*b0 + a1*b1 (+ a2*b2), left to right */
static double dotd(int d, const double* a, const double* b) {
    double r = a[0] * b[0];
    for (int i = 1; i < d; ++i) r = r + a[i] * b[i];
    return r;
}

/* y = M * x  (gemv = axcpy over columns: y_i = (M_i0 x_0) + M_i1 x_1 + ...) ; M d x d column-major */
static void matvec(int d, const double* M, const double* x, double* y) {
    for (int i = 0; i < d; ++i) {
        double r = M[CM(i, 0, d)] * x[0];
        for (int k = 1; k < d; ++k) r = r + M[CM(i, k, d)] * x[k];
        y[i] = r;
    }
}

/* C (m x n) = A (m x k) * B (k x n), all column-major, sequential k accumulation (nalgebra gemm
 * fallback: column-by-column gemv, each an axcpy chain) */
static void matmul(int m, int k, int n, const double* A, const double* B, double* C) {
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) {
            double r = A[CM(i, 0, m)] * B[CM(0, j, k)];
            for (int l = 1; l < k; ++l) r = r + A[CM(i, l, m)] * B[CM(l, j, k)];
            C[CM(i, j, m)] = r;
        }
}

static void transpose(int d, const double* A, double* At) {
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) At[CM(j, i, d)] = A[CM(i, j, d)];
}

/* nalgebra determinant() for 1x1, 2x2, 3x3 */
static double det(int d, const double* m) {
    if (d == 1) return m[0];
    if (d == 2) return m[CM(0, 0, 2)] * m[CM(1, 1, 2)] - m[CM(1, 0, 2)] * m[CM(0, 1, 2)];
    double m11 = m[CM(0, 0, 3)], m12 = m[CM(0, 1, 3)], m13 = m[CM(0, 2, 3)];
    double m21 = m[CM(1, 0, 3)], m22 = m[CM(1, 1, 3)], m23 = m[CM(1, 2, 3)];
    double m31 = m[CM(2, 0, 3)], m32 = m[CM(2, 1, 3)], m33 = m[CM(2, 2, 3)];
    double minor_m12_m23 = m22 * m33 - m32 * m23;
    double minor_m11_m23 = m21 * m33 - m31 * m23;
    double minor_m11_m22 = m21 * m32 - m31 * m22;
    return m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
}

/* nalgebra try_inverse(): cofactors / determinant; fails only if determinant == 0 exactly */
static int try_inverse(int d, const double* m, double* out) {
    if (d == 1) {
        if (m[0] == 0.0) return 0;
        out[0] = 1.0 / m[0];
        return 1;
    }
    if (d == 2) {
        double m11 = m[CM(0, 0, 2)], m12 = m[CM(0, 1, 2)], m21 = m[CM(1, 0, 2)], m22 = m[CM(1, 1, 2)];
        double determinant = m11 * m22 - m21 * m12;
        if (determinant == 0.0) return 0;
        out[CM(0, 0, 2)] = m22 / determinant;
        out[CM(0, 1, 2)] = -m12 / determinant;
        out[CM(1, 0, 2)] = -m21 / determinant;
        out[CM(1, 1, 2)] = m11 / determinant;
        return 1;
    }
    double m11 = m[CM(0, 0, 3)], m12 = m[CM(0, 1, 3)], m13 = m[CM(0, 2, 3)];
    double m21 = m[CM(1, 0, 3)], m22 = m[CM(1, 1, 3)], m23 = m[CM(1, 2, 3)];
    double m31 = m[CM(2, 0, 3)], m32 = m[CM(2, 1, 3)], m33 = m[CM(2, 2, 3)];
    double minor_m12_m23 = m22 * m33 - m32 * m23;
    double minor_m11_m23 = m21 * m33 - m31 * m23;
    double minor_m11_m22 = m21 * m32 - m31 * m22;
    double determinant = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
    if (determinant == 0.0) return 0;
    out[CM(0, 0, 3)] = minor_m12_m23 / determinant;
    out[CM(0, 1, 3)] = (m13 * m32 - m33 * m12) / determinant;
    out[CM(0, 2, 3)] = (m12 * m23 - m22 * m13) / determinant;
    out[CM(1, 0, 3)] = -minor_m11_m23 / determinant;
    out[CM(1, 1, 3)] = (m11 * m33 - m31 * m13) / determinant;
    out[CM(1, 2, 3)] = (m13 * m21 - m23 * m11) / determinant;
    out[CM(2, 0, 3)] = minor_m11_m22 / determinant;
    out[CM(2, 1, 3)] = (m12 * m31 - m32 * m11) / determinant;
    out[CM(2, 2, 3)] = (m11 * m22 - m21 * m12) / determinant;
    return 1;
}

static double trace(int d, const double* m) {
    double r = m[0];
    for (int i = 1; i < d; ++i) r = r + m[CM(i, i, d)];
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * elements
 * ---------------------------------------------------------------------------------------------- */

int fo_element_num_nodes(int k) {
    switch (k) {
        case FO_QUAD4: return 4;
        case FO_HEX8: return 8;
        case FO_TET4: return 4;
        case FO_HEX27: return 27;
        case FO_TRI3: return 3;
        case FO_TET10: return 10;
        case FO_QUAD9: return 9;
        case FO_TRI6: return 6;
        case FO_HEX20: return 20;
        case FO_TET20: return 20;
        default: return -1;
    }
}
int fo_element_dim(int k) {
    switch (k) {
        case FO_QUAD4: case FO_TRI3: case FO_QUAD9: case FO_TRI6: return 2;
        case FO_HEX8: case FO_TET4: case FO_HEX27: case FO_TET10: case FO_HEX20: case FO_TET20: return 3;
        default: return -1;
    }
}
int fo_operator_solution_dim(int op, int d) { return (op == FO_LAPLACE || op == FO_MASS_SCALAR) ? 1 : d; }   /* (FO_TENSOR: s = d) */

/* src/element.rs:244-298 */
static double phi_linear_1d(double alpha, double xi) { return (1.0 + alpha * xi) / 2.0; }
static double phi_linear_1d_grad(double alpha) { return alpha / 2.0; }
static double phi_quadratic_1d(double alpha, double xi) {
    double alpha2 = alpha * alpha;
    double xi2 = xi * xi;
    return (3.0 / 2.0 * alpha2 - 1.0) * xi2 + 0.5 * alpha * xi + 1.0 - alpha2;
}
static double phi_quadratic_1d_grad(double alpha, double xi) {
    double alpha2 = alpha * alpha;
    return 2.0 * (3.0 / 2.0 * alpha2 - 1.0) * xi + 0.5 * alpha;
}

/* node sign tables: src/element/quadrilateral.rs:84-89, hexahedron.rs:49-58, :229-264 */
static const double QUAD4_SIGNS[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
/* quadrilateral.rs:212-228: corners, edge midpoints (0,1) (1,2) (2,3) (3,0), centre */
static const double QUAD9_SIGNS[9][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}, {0, -1}, {1, 0}, {0, 1}, {-1, 0}, {0, 0}};
static const double HEX27_SIGNS[27][3] = {
    {-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1},
    /* edge nodes */
    {0, -1, -1}, {-1, 0, -1}, {-1, -1, 0}, {1, 0, -1}, {1, -1, 0}, {0, 1, -1}, {1, 1, 0}, {-1, 1, 0},
    {0, -1, 1}, {-1, 0, 1}, {1, 0, 1}, {0, 1, 1},
    /* face nodes */
    {0, 0, -1}, {0, -1, 0}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1},
    /* centre */
    {0, 0, 0}};

int fo_element_basis(int kind, const double* xi, double* phi) {
    switch (kind) {
        case FO_QUAD4: /* quadrilateral.rs:79-90 */
            for (int n = 0; n < 4; ++n) {
                double alpha = QUAD4_SIGNS[n][0], beta = QUAD4_SIGNS[n][1];
                phi[n] = (1.0 + alpha * xi[0]) * (1.0 + beta * xi[1]) / 4.0;
            }
            return FO_OK;
        case FO_HEX8: /* hexahedron.rs:43-59 */
            for (int n = 0; n < 8; ++n) {
                const double* s = HEX27_SIGNS[n];
                phi[n] = phi_linear_1d(s[0], xi[0]) * phi_linear_1d(s[1], xi[1]) * phi_linear_1d(s[2], xi[2]);
            }
            return FO_OK;
        case FO_HEX27: /* hexahedron.rs:222-265 */
            for (int n = 0; n < 27; ++n) {
                const double* s = HEX27_SIGNS[n];
                phi[n] = phi_quadratic_1d(s[0], xi[0]) * phi_quadratic_1d(s[1], xi[1]) * phi_quadratic_1d(s[2], xi[2]);
            }
            return FO_OK;
        case FO_TET4: /* tetrahedron.rs:551-558 */
            phi[0] = -0.5 * xi[0] - 0.5 * xi[1] - 0.5 * xi[2] - 0.5;
            phi[1] = 0.5 * xi[0] + 0.5;
            phi[2] = 0.5 * xi[1] + 0.5;
            phi[3] = 0.5 * xi[2] + 0.5;
            return FO_OK;
        case FO_TRI3: /* triangle.rs:72-78 */
            phi[0] = -0.5 * xi[0] - 0.5 * xi[1];
            phi[1] = 0.5 * xi[0] + 0.5;
            phi[2] = 0.5 * xi[1] + 0.5;
            return FO_OK;
        case FO_TET10: { /* tetrahedron.rs:179-195: products of the Tet4 basis */
            double psi[4];
            fo_element_basis(FO_TET4, xi, psi);
            phi[0] = psi[0] * (2.0 * psi[0] - 1.0);
            phi[1] = psi[1] * (2.0 * psi[1] - 1.0);
            phi[2] = psi[2] * (2.0 * psi[2] - 1.0);
            phi[3] = psi[3] * (2.0 * psi[3] - 1.0);
            phi[4] = 4.0 * psi[0] * psi[1];
            phi[5] = 4.0 * psi[1] * psi[2];
            phi[6] = 4.0 * psi[0] * psi[2];
            phi[7] = 4.0 * psi[0] * psi[3];
            phi[8] = 4.0 * psi[2] * psi[3];
            phi[9] = 4.0 * psi[1] * psi[3];
            return FO_OK;
        }
        case FO_TRI6: { /* triangle.rs:211-224 */
            double psi[3];
            fo_element_basis(FO_TRI3, xi, psi);
            phi[0] = psi[0] * (2.0 * psi[0] - 1.0);
            phi[1] = psi[1] * (2.0 * psi[1] - 1.0);
            phi[2] = psi[2] * (2.0 * psi[2] - 1.0);
            phi[3] = 4.0 * psi[0] * psi[1];
            phi[4] = 4.0 * psi[1] * psi[2];
            phi[5] = 4.0 * psi[0] * psi[2];
            return FO_OK;
        }
        case FO_TET20: { /* tetrahedron.rs:346-401 (Zienkiewicz): products of the Tet4 basis */
            static const int ED[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
            static const int FA[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
            double psi[4];
            fo_element_basis(FO_TET4, xi, psi);
            for (int i = 0; i < 4; ++i) phi[i] = 0.5 * psi[i] * (3.0 * psi[i] - 1.0) * (3.0 * psi[i] - 2.0);
            for (int m = 0; m < 6; ++m) {
                int a = ED[m][0], b = ED[m][1];
                phi[4 + 2 * m] = (9.0 / 2.0) * psi[a] * psi[b] * (3.0 * psi[a] - 1.0);      /* phi_edge(a, b) */
                phi[4 + 2 * m + 1] = (9.0 / 2.0) * psi[b] * psi[a] * (3.0 * psi[b] - 1.0);  /* phi_edge(b, a) */
            }
            for (int f = 0; f < 4; ++f) phi[16 + f] = 27.0 * psi[FA[f][0]] * psi[FA[f][1]] * psi[FA[f][2]];
            return FO_OK;
        }
        case FO_HEX20: /* hexahedron.rs:413-462: corner and edge functions; nodes = the first 20 of Hex27 */
            for (int n = 0; n < 20; ++n) {
                double alpha = HEX27_SIGNS[n][0], beta = HEX27_SIGNS[n][1], gamma = HEX27_SIGNS[n][2];
                if (n < 8) {
                    phi[n] = (1.0 / 8.0) * (1.0 + alpha * xi[0]) * (1.0 + beta * xi[1]) * (1.0 + gamma * xi[2]) *
                             (alpha * xi[0] + beta * xi[1] + gamma * xi[2] - 2.0);
                } else {
                    double alpha2 = alpha * alpha, beta2 = beta * beta, gamma2 = gamma * gamma;
                    phi[n] = (1.0 / 4.0) * (1.0 - (1.0 - alpha2) * xi[0] * xi[0]) * (1.0 - (1.0 - beta2) * xi[1] * xi[1]) *
                             (1.0 - (1.0 - gamma2) * xi[2] * xi[2]) * (1.0 + alpha * xi[0]) * (1.0 + beta * xi[1]) *
                             (1.0 + gamma * xi[2]);
                }
            }
            return FO_OK;
        case FO_QUAD9: /* quadrilateral.rs:233-277: N_ab(xi, eta) = phi_a(xi) phi_b(eta), 1-D quadratic factors */
            for (int n = 0; n < 9; ++n)
                phi[n] = phi_quadratic_1d(QUAD9_SIGNS[n][0], xi[0]) * phi_quadratic_1d(QUAD9_SIGNS[n][1], xi[1]);
            return FO_OK;
        default: return FO_BAD_ARGUMENT;
    }
}

int fo_element_gradients(int kind, const double* xi, double* g) {
    switch (kind) {
        case FO_QUAD4: /* quadrilateral.rs:94-107 */
            for (int n = 0; n < 4; ++n) {
                double alpha = QUAD4_SIGNS[n][0], beta = QUAD4_SIGNS[n][1];
                g[CM(0, n, 2)] = alpha * (1.0 + beta * xi[1]) / 4.0;
                g[CM(1, n, 2)] = beta * (1.0 + alpha * xi[0]) / 4.0;
            }
            return FO_OK;
        case FO_HEX8: /* hexahedron.rs:63-83 */
            for (int n = 0; n < 8; ++n) {
                double a = HEX27_SIGNS[n][0], b = HEX27_SIGNS[n][1], c = HEX27_SIGNS[n][2];
                g[CM(0, n, 3)] = phi_linear_1d_grad(a) * phi_linear_1d(b, xi[1]) * phi_linear_1d(c, xi[2]);
                g[CM(1, n, 3)] = phi_linear_1d(a, xi[0]) * phi_linear_1d_grad(b) * phi_linear_1d(c, xi[2]);
                g[CM(2, n, 3)] = phi_linear_1d(a, xi[0]) * phi_linear_1d(b, xi[1]) * phi_linear_1d_grad(c);
            }
            return FO_OK;
        case FO_HEX27: /* hexahedron.rs:269-315 */
            for (int n = 0; n < 27; ++n) {
                double a = HEX27_SIGNS[n][0], b = HEX27_SIGNS[n][1], c = HEX27_SIGNS[n][2];
                g[CM(0, n, 3)] = phi_quadratic_1d_grad(a, xi[0]) * phi_quadratic_1d(b, xi[1]) * phi_quadratic_1d(c, xi[2]);
                g[CM(1, n, 3)] = phi_quadratic_1d(a, xi[0]) * phi_quadratic_1d_grad(b, xi[1]) * phi_quadratic_1d(c, xi[2]);
                g[CM(2, n, 3)] = phi_quadratic_1d(a, xi[0]) * phi_quadratic_1d(b, xi[1]) * phi_quadratic_1d_grad(c, xi[2]);
            }
            return FO_OK;
        case FO_TET4: { /* tetrahedron.rs:561-568 */
            static const double G[12] = {-0.5, -0.5, -0.5, 0.5, 0.0, 0.0, 0.0, 0.5, 0.0, 0.0, 0.0, 0.5};
            memcpy(g, G, sizeof G);
            return FO_OK;
        }
        case FO_TRI3: { /* triangle.rs:82-89 */
            static const double G[6] = {-0.5, -0.5, 0.5, 0.0, 0.0, 0.5};
            memcpy(g, G, sizeof G);
            return FO_OK;
        }
        case FO_TET10:
        case FO_TRI6: {
            /* tetrahedron.rs:198-224, triangle.rs:228-252: vertex node i: g_i (4 psi_i - 1);
             * edge node (i, j): g_i (4 psi_j) + g_j (4 psi_i) */
            static const int E3[6][2] = {{0, 1}, {1, 2}, {0, 2}, {0, 3}, {2, 3}, {1, 3}};
            static const int E2[3][2] = {{0, 1}, {1, 2}, {0, 2}};
            int lin = (kind == FO_TET10) ? FO_TET4 : FO_TRI3, d = (kind == FO_TET10) ? 3 : 2, nv = d + 1;
            int ne = (kind == FO_TET10) ? 6 : 3;
            double psi[4], gl[12];
            fo_element_basis(lin, xi, psi);
            fo_element_gradients(lin, xi, gl);
            for (int i = 0; i < nv; ++i)
                for (int k = 0; k < d; ++k) g[CM(k, i, d)] = gl[CM(k, i, d)] * (4.0 * psi[i] - 1.0);
            for (int m = 0; m < ne; ++m) {
                int i = (kind == FO_TET10) ? E3[m][0] : E2[m][0], j = (kind == FO_TET10) ? E3[m][1] : E2[m][1];
                for (int k = 0; k < d; ++k)
                    g[CM(k, nv + m, d)] = gl[CM(k, i, d)] * (4.0 * psi[j]) + gl[CM(k, j, d)] * (4.0 * psi[i]);
            }
            return FO_OK;
        }
        case FO_TET20: { /* tetrahedron.rs:404-466 */
            static const int ED[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
            static const int FA[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
            double psi[4], gl[12];
            fo_element_basis(FO_TET4, xi, psi);
            fo_element_gradients(FO_TET4, xi, gl);
            for (int i = 0; i < 4; ++i) {
                double pp = psi[i];
                for (int k = 0; k < 3; ++k) g[CM(k, i, 3)] = gl[CM(k, i, 3)] * 0.5 * (27.0 * pp * pp - 18.0 * pp + 2.0);
            }
            for (int m = 0; m < 6; ++m)
                for (int half = 0; half < 2; ++half) {
                    int a = half ? ED[m][1] : ED[m][0], b = half ? ED[m][0] : ED[m][1];
                    double pa = psi[a], pb = psi[b];
                    for (int k = 0; k < 3; ++k)
                        g[CM(k, 4 + 2 * m + half, 3)] =
                            (gl[CM(k, a, 3)] * (pb * (6.0 * pa - 1.0)) + gl[CM(k, b, 3)] * (pa * (3.0 * pa - 1.0))) * (9.0 / 2.0);
                }
            for (int f = 0; f < 4; ++f) {
                int a = FA[f][0], b = FA[f][1], c = FA[f][2];
                for (int k = 0; k < 3; ++k)
                    g[CM(k, 16 + f, 3)] = (gl[CM(k, a, 3)] * psi[b] * psi[c] + gl[CM(k, b, 3)] * psi[a] * psi[c] +
                                           gl[CM(k, c, 3)] * psi[a] * psi[b]) * 27.0;
            }
            return FO_OK;
        }
        case FO_HEX20: /* hexahedron.rs:465-543 */
            for (int n = 0; n < 20; ++n) {
                double alpha = HEX27_SIGNS[n][0], beta = HEX27_SIGNS[n][1], gamma = HEX27_SIGNS[n][2];
                double gg = (1.0 + alpha * xi[0]) * (1.0 + beta * xi[1]) * (1.0 + gamma * xi[2]);
                if (n < 8) {
                    double f = alpha * xi[0] + beta * xi[1] + gamma * xi[2] - 2.0, sc = 1.0 / 8.0;
                    g[CM(0, n, 3)] = sc * (alpha * gg + f * alpha * (1.0 + beta * xi[1]) * (1.0 + gamma * xi[2]));
                    g[CM(1, n, 3)] = sc * (beta * gg + f * beta * (1.0 + alpha * xi[0]) * (1.0 + gamma * xi[2]));
                    g[CM(2, n, 3)] = sc * (gamma * gg + f * gamma * (1.0 + alpha * xi[0]) * (1.0 + beta * xi[1]));
                } else {
                    double alpha2 = alpha * alpha, beta2 = beta * beta, gamma2 = gamma * gamma, sc = 1.0 / 4.0;
                    double h = (1.0 - (1.0 - alpha2) * xi[0] * xi[0]) * (1.0 - (1.0 - beta2) * xi[1] * xi[1]) *
                               (1.0 - (1.0 - gamma2) * xi[2] * xi[2]);
                    double dh0 = -2.0 * (1.0 - alpha2) * xi[0] * (1.0 - (1.0 - beta2) * xi[1] * xi[1]) * (1.0 - (1.0 - gamma2) * xi[2] * xi[2]);
                    double dh1 = -2.0 * (1.0 - beta2) * xi[1] * (1.0 - (1.0 - alpha2) * xi[0] * xi[0]) * (1.0 - (1.0 - gamma2) * xi[2] * xi[2]);
                    double dh2 = -2.0 * (1.0 - gamma2) * xi[2] * (1.0 - (1.0 - alpha2) * xi[0] * xi[0]) * (1.0 - (1.0 - beta2) * xi[1] * xi[1]);
                    g[CM(0, n, 3)] = sc * (dh0 * gg + h * alpha * (1.0 + beta * xi[1]) * (1.0 + gamma * xi[2]));
                    g[CM(1, n, 3)] = sc * (dh1 * gg + h * beta * (1.0 + alpha * xi[0]) * (1.0 + gamma * xi[2]));
                    g[CM(2, n, 3)] = sc * (dh2 * gg + h * gamma * (1.0 + alpha * xi[0]) * (1.0 + beta * xi[1]));
                }
            }
            return FO_OK;
        case FO_QUAD9: /* quadrilateral.rs:280-313 */
            for (int n = 0; n < 9; ++n) {
                double alpha = QUAD9_SIGNS[n][0], beta = QUAD9_SIGNS[n][1];
                g[CM(0, n, 2)] = phi_quadratic_1d(beta, xi[1]) * phi_quadratic_1d_grad(alpha, xi[0]);
                g[CM(1, n, 2)] = phi_quadratic_1d(alpha, xi[0]) * phi_quadratic_1d_grad(beta, xi[1]);
            }
            return FO_OK;
        default: return FO_BAD_ARGUMENT;
    }
}

/* J = X * G^T, X[i][j] = vertex_j[i]  (hexahedron.rs:101-107, tetrahedron.rs:584-590,
 * quadrilateral.rs:125-132).  Hex27 delegates to its embedded Hex8 (hexahedron.rs:324-326). */
/* the quadratic elements take their geometry from the embedded linear element (hexahedron.rs:324-326,
 * tetrahedron.rs:233-240, quadrilateral.rs:316-323, triangle.rs:254-262) */
static int geometry_kind(int kind) {
    switch (kind) {
        case FO_HEX27: return FO_HEX8;
        case FO_HEX20: return FO_HEX8;
        case FO_TET10: return FO_TET4;
        case FO_TET20: return FO_TET4;
        case FO_QUAD9: return FO_QUAD4;
        case FO_TRI6: return FO_TRI3;
        default: return kind;
    }
}

int fo_element_reference_jacobian(int kind, const double* ev, const double* xi, double* J) {
    int gkind = geometry_kind(kind);
    int n = fo_element_num_nodes(gkind), d = fo_element_dim(gkind);
    if (n < 0) return FO_BAD_ARGUMENT;
    double G[MAXD * 8];
    fo_element_gradients(gkind, xi, G);
    /* (X * Gt)[i][j] = sum_k X[i][k] * Gt[k][j] = sum_k v_k[i] * G[j][k], k ascending */
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) {
            double r = ev[0 * d + i] * G[CM(j, 0, d)];
            for (int k = 1; k < n; ++k) r = r + ev[k * d + i] * G[CM(j, k, d)];
            J[CM(i, j, d)] = r;
        }
    return FO_OK;
}

/* map_reference_coords: x = X * N^T (hexahedron.rs:92-98) ; Hex27 uses the Hex8 map */
static void map_reference_coords(int kind, const double* ev, const double* xi, double* x) {
    int gkind = geometry_kind(kind);
    int n = fo_element_num_nodes(gkind), d = fo_element_dim(gkind);
    double N[8];
    fo_element_basis(gkind, xi, N);
    for (int i = 0; i < d; ++i) {
        double r = ev[0 * d + i] * N[0];
        for (int k = 1; k < n; ++k) r = r + ev[k * d + i] * N[k];
        x[i] = r;
    }
}

/* ------------------------------------------------------------------------------------------------
 * quadrature
 * ---------------------------------------------------------------------------------------------- */

/* LegendreRecurrence fenris-quadrature/src/univariate.rs:13-55 */
static void legendre(int n, double x, double* p, double* dp) {
    double p1 = 1.0, p2 = 0.0, p3;
    for (int mi = 1; mi <= n; ++mi) {
        double m = (double)mi;
        p3 = p2;
        p2 = p1;
        p1 = ((2.0 * m - 1.0) * x * p2 - (m - 1.0) * p3) / m;
    }
    double nn = (double)n;
    *p = p1;
    *dp = nn * (x * p1 - p2) / (x * x - 1.0);
}

/* univariate.rs:66-118 */
int fo_gauss(int n, double* weights, double* points) {
    if (n <= 0) return FO_BAD_ARGUMENT;
    const double PI = 3.14159265358979323846264338327950288;
    int m = (n + 1) / 2;
    for (int i = 0; i < m; ++i) {
        double x = cos(PI * ((double)i + 0.75) / ((double)n + 0.5));
        double p, dp;
        legendre(n, x, &p, &dp);
        for (;;) {
            double dx = -p / dp;
            x += dx;
            legendre(n, x, &p, &dp);
            if (fabs(dx) <= 1e-15) break;
        }
        double w = 2.0 / ((1.0 - x * x) * dp * dp);
        points[i] = x;
        weights[i] = w;
    }
    for (int i = m; i < n; ++i) {
        int mirror = n - i - 1;
        points[i] = -points[mirror];
        weights[i] = weights[mirror];
    }
    return FO_OK;
}

/* tensor.rs:13-31 */
int fo_quadrilateral_gauss(int n, double* w2, double* p2) {
    double w1[256], x1[256];
    if (n <= 0 || n > 256) return FO_BAD_ARGUMENT;
    fo_gauss(n, w1, x1);
    size_t c = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            w2[c] = w1[i] * w1[j];
            p2[2 * c + 0] = x1[i];
            p2[2 * c + 1] = x1[j];
            ++c;
        }
    return FO_OK;
}

/* tensor.rs:36-55 */
int fo_hexahedron_gauss(int n, double* w3, double* p3) {
    double w1[256], x1[256];
    if (n <= 0 || n > 256) return FO_BAD_ARGUMENT;
    fo_gauss(n, w1, x1);
    size_t c = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                w3[c] = w1[i] * w1[j] * w1[k];
                p3[3 * c + 0] = x1[i];
                p3[3 * c + 1] = x1[j];
                p3[3 * c + 2] = x1[k];
                ++c;
            }
    return FO_OK;
}

/* Witherden-Vincent tables, fenris-quadrature/rules/polyquad/expanded/{tet,tri}/ *.txt (data, generated into
 * polyquad_tables.inc), parsed like Rust str::parse::<f64> (correctly rounded) -> strtod.
 * select_minimum (build.rs:172-194): the smallest tabulated strength >= the requested one. */
#include "polyquad_tables.inc"
int fo_tetrahedron_rule(int strength, double* w, double* p) {
    for (size_t t = 0; t < sizeof FO_PQ_TET_TABLES / sizeof FO_PQ_TET_TABLES[0]; ++t) {
        if ((int)FO_PQ_TET_TABLES[t].strength < strength) continue;
        int np = (int)FO_PQ_TET_TABLES[t].npts;
        for (int i = 0; i < np; ++i) {
            for (int c = 0; c < 3; ++c) p[3 * i + c] = strtod(FO_PQ_TET_TABLES[t].rows[4 * i + c], NULL);
            w[i] = strtod(FO_PQ_TET_TABLES[t].rows[4 * i + 3], NULL);
        }
        return np;
    }
    return -1;
}

int fo_triangle_rule(int strength, double* w, double* p) {
    for (size_t t = 0; t < sizeof FO_PQ_TRI_TABLES / sizeof FO_PQ_TRI_TABLES[0]; ++t) {
        if ((int)FO_PQ_TRI_TABLES[t].strength < strength) continue;
        int np = (int)FO_PQ_TRI_TABLES[t].npts;
        for (int i = 0; i < np; ++i) {
            for (int c = 0; c < 2; ++c) p[2 * i + c] = strtod(FO_PQ_TRI_TABLES[t].rows[3 * i + c], NULL);
            w[i] = strtod(FO_PQ_TRI_TABLES[t].rows[3 * i + 2], NULL);
        }
        return np;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------------
 * mesh generators
 * ---------------------------------------------------------------------------------------------- */

void fo_free(void* p) { free(p); }

/* src/mesh/procedural.rs:46-93 */
int fo_create_rectangular_uniform_quad_mesh_2d(double unit_length, uint64_t units_x, uint64_t units_y,
                                               uint64_t cells_per_unit, const double top_left[2], double** vertices,
                                               uint64_t* num_vertices, uint64_t** connectivity, uint64_t* num_cells) {
    *vertices = NULL; *connectivity = NULL; *num_vertices = 0; *num_cells = 0;
    if (cells_per_unit == 0 || units_x == 0 || units_y == 0) return FO_OK;
    double cell_size = unit_length / (double)cells_per_unit;
    uint64_t ncx = units_x * cells_per_unit, ncy = units_y * cells_per_unit;
    uint64_t nvx = ncx + 1, nvy = ncy + 1;
    double* v = malloc(sizeof(double) * 2 * nvx * nvy);
    uint64_t* c = malloc(sizeof(uint64_t) * 4 * ncx * ncy);
    if (!v || !c) return FO_BAD_ARGUMENT;
    size_t k = 0;
    for (uint64_t j = 0; j < nvy; ++j)
        for (uint64_t i = 0; i < nvx; ++i) {
            /* v = top_left + Vector2(i, -j) * cell_size */
            v[k++] = top_left[0] + (double)i * cell_size;
            v[k++] = top_left[1] + (-(double)j) * cell_size;
        }
#define QIDX(i, j) ((ncx + 1) * (j) + (i))
    k = 0;
    for (uint64_t j = 0; j < ncy; ++j)
        for (uint64_t i = 0; i < ncx; ++i) {
            c[k++] = QIDX(i, j + 1);
            c[k++] = QIDX(i + 1, j + 1);
            c[k++] = QIDX(i + 1, j);
            c[k++] = QIDX(i, j);
        }
#undef QIDX
    *vertices = v; *connectivity = c; *num_vertices = nvx * nvy; *num_cells = ncx * ncy;
    return FO_OK;
}

/* src/mesh/procedural.rs:216-277 */
int fo_create_rectangular_uniform_hex_mesh(double unit_length, uint64_t units_x, uint64_t units_y, uint64_t units_z,
                                           uint64_t cells_per_unit, double** vertices, uint64_t* num_vertices,
                                           uint64_t** connectivity, uint64_t* num_cells) {
    *vertices = NULL; *connectivity = NULL; *num_vertices = 0; *num_cells = 0;
    if (cells_per_unit == 0 || units_x == 0 || units_y == 0) return FO_OK;
    double cell_size = unit_length / (double)cells_per_unit;
    uint64_t ncx = units_x * cells_per_unit, ncy = units_y * cells_per_unit, ncz = units_z * cells_per_unit;
    uint64_t nvx = ncx + 1, nvy = ncy + 1, nvz = ncz + 1;
    double* v = malloc(sizeof(double) * 3 * nvx * nvy * nvz + 8);
    uint64_t* c = malloc(sizeof(uint64_t) * 8 * ncx * ncy * ncz + 8);
    if (!v || !c) return FO_BAD_ARGUMENT;
    size_t q = 0;
    for (uint64_t k = 0; k < nvz; ++k)
        for (uint64_t j = 0; j < nvy; ++j)
            for (uint64_t i = 0; i < nvx; ++i) {
                v[q++] = (double)i * cell_size;
                v[q++] = (double)j * cell_size;
                v[q++] = (double)k * cell_size;
            }
#define HIDX(i, j, k) ((nvx * nvy) * (k) + nvx * (j) + (i))
    q = 0;
    for (uint64_t k = 0; k < ncz; ++k)
        for (uint64_t j = 0; j < ncy; ++j)
            for (uint64_t i = 0; i < ncx; ++i) {
                c[q++] = HIDX(i, j, k);
                c[q++] = HIDX(i + 1, j, k);
                c[q++] = HIDX(i + 1, j + 1, k);
                c[q++] = HIDX(i, j + 1, k);
                c[q++] = HIDX(i, j, k + 1);
                c[q++] = HIDX(i + 1, j, k + 1);
                c[q++] = HIDX(i + 1, j + 1, k + 1);
                c[q++] = HIDX(i, j + 1, k + 1);
            }
#undef HIDX
    *vertices = v; *connectivity = c; *num_vertices = nvx * nvy * nvz; *num_cells = ncx * ncy * ncz;
    return FO_OK;
}

/* src/mesh/procedural.rs:286-403 (BCC lattice) */
typedef struct { uint64_t* data; size_t len, cap; } u64vec;
static void u64vec_push4(u64vec* v, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    if (v->len + 4 > v->cap) {
        v->cap = v->cap ? 2 * v->cap : 1024;
        v->data = realloc(v->data, v->cap * sizeof(uint64_t));
    }
    v->data[v->len++] = a; v->data[v->len++] = b; v->data[v->len++] = c; v->data[v->len++] = d;
}

int fo_create_rectangular_uniform_tet_mesh(double unit_length, uint64_t units_x, uint64_t units_y, uint64_t units_z,
                                           uint64_t cells_per_unit, double** vertices, uint64_t* num_vertices,
                                           uint64_t** connectivity, uint64_t* num_cells) {
    *vertices = NULL; *connectivity = NULL; *num_vertices = 0; *num_cells = 0;
    if (units_x == 0 || units_y == 0 || units_z == 0 || cells_per_unit == 0) return FO_OK;
    double cell_size = unit_length / (double)cells_per_unit;
    uint64_t cx = units_x * cells_per_unit, cy = units_y * cells_per_unit, cz = units_z * cells_per_unit;
    uint64_t vx = cx + 1, vy = cy + 1, vz = cz + 1;
    uint64_t nv = vx * vy * vz + cx * cy * cz;
    double* v = malloc(sizeof(double) * 3 * nv);
    if (!v) return FO_BAD_ARGUMENT;
    size_t q = 0;
    for (uint64_t k = 0; k < vz; ++k)
        for (uint64_t j = 0; j < vy; ++j)
            for (uint64_t i = 0; i < vx; ++i) {
                v[q++] = cell_size * (double)i;
                v[q++] = cell_size * (double)j;
                v[q++] = cell_size * (double)k;
            }
    uint64_t cell_center_offset = vx * vy * vz;
    for (uint64_t k = 0; k < cz; ++k)
        for (uint64_t j = 0; j < cy; ++j)
            for (uint64_t i = 0; i < cx; ++i) {
                v[q++] = cell_size * (0.5 + (double)i);
                v[q++] = cell_size * (0.5 + (double)j);
                v[q++] = cell_size * (0.5 + (double)k);
            }
#define VIDX(i, j, k) ((vx * vy) * (k) + vx * (j) + (i))
#define CIDX(i, j, k) ((cx * cy) * (k) + cx * (j) + (i) + cell_center_offset)
    static const uint64_t FACE[3][4][3] = {{{1, 0, 1}, {1, 1, 1}, {1, 1, 0}, {1, 0, 0}},
                                           {{0, 1, 0}, {1, 1, 0}, {1, 1, 1}, {0, 1, 1}},
                                           {{0, 1, 1}, {1, 1, 1}, {1, 0, 1}, {0, 0, 1}}};
    u64vec conn = {0};
    for (uint64_t k = 0; k < cz; ++k)
        for (uint64_t j = 0; j < cy; ++j)
            for (uint64_t i = 0; i < cx; ++i) {
                uint64_t cell[3] = {i, j, k};
                uint64_t ncell[3] = {cx, cy, cz};
                for (int axis = 0; axis < 3; ++axis) {
                    if (cell[axis] + 1 < ncell[axis]) {
                        /* connect_centers_with_tets */
                        uint64_t fv[4];
                        for (int f = 0; f < 4; ++f)
                            fv[f] = VIDX(i + FACE[axis][f][0], j + FACE[axis][f][1], k + FACE[axis][f][2]);
                        uint64_t c1 = CIDX(i, j, k);
                        uint64_t c2 = CIDX(i + (axis == 0), j + (axis == 1), k + (axis == 2));
                        for (int f = 0; f < 4; ++f) {
                            uint64_t v1 = fv[f], v2 = fv[(f + 1) % 4];
                            u64vec_push4(&conn, c1, c2, v2, v1);
                        }
                    }
                    for (int pass = 0; pass < 2; ++pass) {
                        int positive_dir = pass;
                        if (pass == 0 && cell[axis] != 0) continue;
                        if (pass == 1 && cell[axis] + 1 != ncell[axis]) continue;
                        /* make_pyramid */
                        int64_t fc[4][3];
                        for (int f = 0; f < 4; ++f) {
                            fc[f][0] = (int64_t)(FACE[axis][f][0] + i);
                            fc[f][1] = (int64_t)(FACE[axis][f][1] + j);
                            fc[f][2] = (int64_t)(FACE[axis][f][2] + k);
                        }
                        if (!positive_dir) {
                            for (int f = 0; f < 2; ++f)
                                for (int t = 0; t < 3; ++t) {
                                    int64_t tmp = fc[f][t]; fc[f][t] = fc[3 - f][t]; fc[3 - f][t] = tmp;
                                }
                            for (int f = 0; f < 4; ++f) fc[f][axis] -= 1;
                        }
                        uint64_t a = VIDX((uint64_t)fc[0][0], (uint64_t)fc[0][1], (uint64_t)fc[0][2]);
                        uint64_t b = VIDX((uint64_t)fc[1][0], (uint64_t)fc[1][1], (uint64_t)fc[1][2]);
                        uint64_t c = VIDX((uint64_t)fc[2][0], (uint64_t)fc[2][1], (uint64_t)fc[2][2]);
                        uint64_t dd = VIDX((uint64_t)fc[3][0], (uint64_t)fc[3][1], (uint64_t)fc[3][2]);
                        uint64_t center = CIDX(i, j, k);
                        if ((i + j + k) % 2 == 0) {
                            u64vec_push4(&conn, a, b, c, center);
                            u64vec_push4(&conn, a, c, dd, center);
                        } else {
                            u64vec_push4(&conn, a, b, dd, center);
                            u64vec_push4(&conn, b, c, dd, center);
                        }
                    }
                }
            }
#undef VIDX
#undef CIDX
    *vertices = v; *num_vertices = nv; *connectivity = conn.data; *num_cells = conn.len / 4;
    return FO_OK;
}

/* Hex8 -> Hex27: src/mesh_convert.rs:85-166 (refine) + :227-330 (first-occurrence relabelling keyed on
 * (child index, sorted parent set)).  Open-addressing hash on the sorted parent tuple. */
typedef struct { uint64_t key[8]; uint8_t nkey; uint64_t value; uint8_t used; } parent_slot;

static uint64_t hash_parents(const uint64_t* k, int n) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
    for (int i = 0; i < n; ++i) { h ^= k[i] + 0x9e3779b97f4a7c15ull; h *= 1099511628211ull; h ^= h >> 29; }
    return h;
}
static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x > y) - (x < y);
}

int fo_hex8_to_hex27(const double* vertices, uint64_t num_vertices, const uint64_t* hex8, uint64_t num_cells,
                     double** out_vertices, uint64_t* out_num_vertices, uint64_t** out_connectivity) {
    (void)num_vertices;
    static const int EDGES[12][2] = {{0, 1}, {0, 3}, {0, 4}, {1, 2}, {1, 5}, {2, 3}, {2, 6}, {3, 7}, {4, 5}, {4, 7}, {5, 6}, {6, 7}};
    static const int FACES[6][4] = {{0, 1, 2, 3}, {0, 1, 4, 5}, {0, 3, 4, 7}, {1, 2, 5, 6}, {2, 3, 6, 7}, {4, 5, 6, 7}};
    static const double FACE_XI[6][3] = {{0, 0, -1}, {0, -1, 0}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    size_t cap = 1;
    while (cap < (size_t)num_cells * 27 * 2 + 16) cap <<= 1;
    parent_slot* table = calloc(cap, sizeof(parent_slot));
    double* fv = malloc(sizeof(double) * 3 * ((size_t)num_cells * 27 + 1));
    uint64_t* conn = malloc(sizeof(uint64_t) * ((size_t)num_cells * 27 + 1));
    if (!table || !fv || !conn) return FO_BAD_ARGUMENT;
    uint64_t next = 0;
    for (uint64_t e = 0; e < num_cells; ++e) {
        const uint64_t* gi = hex8 + 8 * e;
        double ev[24];
        for (int n = 0; n < 8; ++n)
            for (int c = 0; c < 3; ++c) ev[3 * n + c] = vertices[3 * gi[n] + c];
        double lv[27][3];
        uint64_t par[27][8];
        int npar[27];
        for (int n = 0; n < 8; ++n) {
            for (int c = 0; c < 3; ++c) lv[n][c] = ev[3 * n + c];
            par[n][0] = gi[n]; npar[n] = 1;
        }
        for (int ed = 0; ed < 12; ++ed) {
            int b = EDGES[ed][0], en = EDGES[ed][1];
            /* nalgebra lerp: self * (1 - t) + rhs * t */
            for (int c = 0; c < 3; ++c) lv[8 + ed][c] = ev[3 * b + c] * (1.0 - 0.5) + ev[3 * en + c] * 0.5;
            par[8 + ed][0] = gi[b]; par[8 + ed][1] = gi[en]; npar[8 + ed] = 2;
        }
        for (int f = 0; f < 6; ++f) {
            map_reference_coords(FO_HEX8, ev, FACE_XI[f], lv[20 + f]);
            for (int t = 0; t < 4; ++t) par[20 + f][t] = gi[FACES[f][t]];
            npar[20 + f] = 4;
        }
        {
            double origin[3] = {0, 0, 0};
            map_reference_coords(FO_HEX8, ev, origin, lv[26]);
            for (int t = 0; t < 8; ++t) par[26][t] = gi[t];
            npar[26] = 8;
        }
        for (int n = 0; n < 27; ++n) {
            qsort(par[n], (size_t)npar[n], sizeof(uint64_t), cmp_u64);
            uint64_t h = hash_parents(par[n], npar[n]) & (cap - 1);
            for (;;) {
                parent_slot* s = &table[h];
                if (!s->used) {
                    s->used = 1; s->nkey = (uint8_t)npar[n];
                    memcpy(s->key, par[n], sizeof(uint64_t) * (size_t)npar[n]);
                    s->value = next;
                    for (int c = 0; c < 3; ++c) fv[3 * next + c] = lv[n][c];
                    conn[27 * e + n] = next++;
                    break;
                }
                if (s->nkey == npar[n] && memcmp(s->key, par[n], sizeof(uint64_t) * (size_t)npar[n]) == 0) {
                    conn[27 * e + n] = s->value;
                    break;
                }
                h = (h + 1) & (cap - 1);
            }
        }
    }
    free(table);
    *out_vertices = fv; *out_num_vertices = next; *out_connectivity = conn;
    return FO_OK;
}

/* Tet20Mesh::from(&tet4_mesh), src/mesh_convert.rs:658-775: every new vertex is a 4-tuple -- [idx,0,0,0] (vertex),
 * [min,max,local,1] (edge node, local counted from min), [a,b,c,2] sorted (face) -- the tuples are sorted and
 * deduplicated, and a vertex is labelled by its rank. */
static int cmp_key4(const void* a, const void* b) {
    const uint64_t* x = a;
    const uint64_t* y = b;
    for (int i = 0; i < 4; ++i)
        if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
}
static void tet20_keys(const uint64_t* g, uint64_t (*k)[4]) {
    static const int ED[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    static const int FA[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
    for (int a = 0; a < 4; ++a) { k[a][0] = g[a]; k[a][1] = 0; k[a][2] = 0; k[a][3] = 0; }
    for (int m = 0; m < 6; ++m)
        for (uint64_t local = 0; local < 2; ++local) {
            uint64_t st = g[ED[m][0]], en = g[ED[m][1]], l = local;
            if (st > en) { uint64_t t = st; st = en; en = t; l = (l + 1) % 2; }
            uint64_t* q = k[4 + 2 * m + (int)local];
            q[0] = st; q[1] = en; q[2] = l; q[3] = 1;
        }
    for (int f = 0; f < 4; ++f) {
        uint64_t t[3] = {g[FA[f][0]], g[FA[f][1]], g[FA[f][2]]};
        qsort(t, 3, sizeof(uint64_t), cmp_u64);
        uint64_t* q = k[16 + f];
        q[0] = t[0]; q[1] = t[1]; q[2] = t[2]; q[3] = 2;
    }
}
int fo_tet4_to_tet20(const double* vertices, uint64_t num_vertices, const uint64_t* tet4, uint64_t num_cells,
                     double** out_vertices, uint64_t* out_num_vertices, uint64_t** out_connectivity) {
    (void)num_vertices;
    size_t total = (size_t)num_cells * 20;
    uint64_t (*all)[4] = malloc(sizeof(uint64_t[4]) * (total + 1));
    uint64_t* conn = malloc(sizeof(uint64_t) * (total + 1));
    if (!all || !conn) return FO_BAD_ARGUMENT;
    for (uint64_t e = 0; e < num_cells; ++e) tet20_keys(tet4 + 4 * e, all + 20 * e);
    qsort(all, total, sizeof(uint64_t[4]), cmp_key4);
    size_t nu = 0;
    for (size_t i = 0; i < total; ++i)
        if (nu == 0 || cmp_key4(all[nu - 1], all[i]) != 0) { memcpy(all[nu], all[i], sizeof(uint64_t[4])); ++nu; }
    for (uint64_t e = 0; e < num_cells; ++e) {
        uint64_t k[20][4];
        tet20_keys(tet4 + 4 * e, k);
        for (int a = 0; a < 20; ++a) {
            size_t lo = 0, hi = nu;  /* binary_search :740-742 */
            while (lo < hi) {
                size_t mid = (lo + hi) / 2;
                if (cmp_key4(all[mid], k[a]) < 0) lo = mid + 1; else hi = mid;
            }
            conn[20 * e + a] = lo;
        }
    }
    double* fv = malloc(sizeof(double) * 3 * (nu + 1));
    for (size_t i = 0; i < nu; ++i)
        for (int c = 0; c < 3; ++c) {
            const uint64_t* q = all[i];
            if (q[3] == 0) fv[3 * i + c] = vertices[3 * q[0] + c];
            else if (q[3] == 1) {
                double st = vertices[3 * q[0] + c], en = vertices[3 * q[1] + c];
                double alpha = (double)(q[2] + 1) / 3.0;
                fv[3 * i + c] = st + (en - st) * alpha;
            } else {
                fv[3 * i + c] = ((vertices[3 * q[0] + c] + vertices[3 * q[1] + c]) + vertices[3 * q[2] + c]) / 3.0;
            }
        }
    free(all);
    *out_vertices = fv; *out_num_vertices = nu; *out_connectivity = conn;
    return FO_OK;
}

/* p-refinement to the quadratic meshes (src/mesh_convert.rs).
 * Tet4 -> Tet10: RefineFrom :42-83 (vertex nodes, then the edge nodes (0,1) (1,2) (0,2) (0,3) (2,3) (1,3) at
 *   lerp(begin, end, 0.5)) through the first-occurrence relabelling of :227-330, like Hex8 -> Hex27.
 * Tri3 -> Tri6 (:332-383), Quad4 -> Quad9 (:385-442): the vertices are kept; for every element the midpoints of its
 *   edges (consecutive vertices, cyclically), keyed by (min, max), are appended at first occurrence as
 *   (v_a + v_b) / 2; Quad9 then appends map_reference_coords(origin) of the cell. */
int fo_refine_to_quadratic(int from_kind, const double* vertices, uint64_t num_vertices, const uint64_t* conn_in,
                           uint64_t num_cells, double** out_vertices, uint64_t* out_num_vertices, uint64_t** out_connectivity) {
    if (from_kind == FO_TET4 || from_kind == FO_HEX8) {
        /* Hex8 -> Hex20 (mesh_convert.rs:168-217): the vertex and edge nodes of the Hex27 refinement */
        static const int EDGES_T[6][2] = {{0, 1}, {1, 2}, {0, 2}, {0, 3}, {2, 3}, {1, 3}};
        static const int EDGES_H[12][2] = {{0, 1}, {0, 3}, {0, 4}, {1, 2}, {1, 5}, {2, 3}, {2, 6}, {3, 7}, {4, 5}, {4, 7}, {5, 6}, {6, 7}};
        const int (*EDGES)[2] = (from_kind == FO_HEX8) ? EDGES_H : EDGES_T;
        const int NV0 = (from_kind == FO_HEX8) ? 8 : 4, NE = (from_kind == FO_HEX8) ? 12 : 6, N1 = NV0 + NE;
        size_t cap = 1;
        while (cap < (size_t)num_cells * 20 * 2 + 16) cap <<= 1;
        parent_slot* table = calloc(cap, sizeof(parent_slot));
        double* fv = malloc(sizeof(double) * 3 * ((size_t)num_cells * 20 + 1));
        uint64_t* conn = malloc(sizeof(uint64_t) * ((size_t)num_cells * 20 + 1));
        if (!table || !fv || !conn) return FO_BAD_ARGUMENT;
        uint64_t next = 0;
        for (uint64_t e = 0; e < num_cells; ++e) {
            const uint64_t* gi = conn_in + (uint64_t)NV0 * e;
            double lv[20][3];
            uint64_t par[20][2];
            int npar[20];
            for (int n = 0; n < NV0; ++n) {
                for (int c = 0; c < 3; ++c) lv[n][c] = vertices[3 * gi[n] + c];
                par[n][0] = gi[n]; npar[n] = 1;
            }
            for (int ed = 0; ed < NE; ++ed) {
                int b = EDGES[ed][0], en = EDGES[ed][1];
                for (int c = 0; c < 3; ++c) lv[NV0 + ed][c] = vertices[3 * gi[b] + c] * (1.0 - 0.5) + vertices[3 * gi[en] + c] * 0.5;
                par[NV0 + ed][0] = gi[b]; par[NV0 + ed][1] = gi[en]; npar[NV0 + ed] = 2;
            }
            for (int n = 0; n < N1; ++n) {
                qsort(par[n], (size_t)npar[n], sizeof(uint64_t), cmp_u64);
                uint64_t h = hash_parents(par[n], npar[n]) & (cap - 1);
                for (;;) {
                    parent_slot* sl = &table[h];
                    if (!sl->used) {
                        sl->used = 1; sl->nkey = (uint8_t)npar[n];
                        memcpy(sl->key, par[n], sizeof(uint64_t) * (size_t)npar[n]);
                        sl->value = next;
                        for (int c = 0; c < 3; ++c) fv[3 * next + c] = lv[n][c];
                        conn[(uint64_t)N1 * e + n] = next++;
                        break;
                    }
                    if (sl->nkey == npar[n] && memcmp(sl->key, par[n], sizeof(uint64_t) * (size_t)npar[n]) == 0) {
                        conn[(uint64_t)N1 * e + n] = sl->value;
                        break;
                    }
                    h = (h + 1) & (cap - 1);
                }
            }
        }
        free(table);
        *out_vertices = fv; *out_num_vertices = next; *out_connectivity = conn;
        return FO_OK;
    }
    if (from_kind != FO_TRI3 && from_kind != FO_QUAD4) return FO_BAD_ARGUMENT;
    int n0 = (from_kind == FO_TRI3) ? 3 : 4, n1 = (from_kind == FO_TRI3) ? 6 : 9;
    size_t cap = 1;
    while (cap < (size_t)num_cells * n0 * 2 + 16) cap <<= 1;
    parent_slot* table = calloc(cap, sizeof(parent_slot));
    double* fv = malloc(sizeof(double) * 2 * ((size_t)num_vertices + (size_t)num_cells * (n0 + 1) + 1));
    uint64_t* conn = malloc(sizeof(uint64_t) * ((size_t)num_cells * n1 + 1));
    if (!table || !fv || !conn) return FO_BAD_ARGUMENT;
    memcpy(fv, vertices, sizeof(double) * 2 * (size_t)num_vertices);
    uint64_t next = num_vertices;
    for (uint64_t e = 0; e < num_cells; ++e) {
        const uint64_t* gi = conn_in + (uint64_t)n0 * e;
        uint64_t* o = conn + (uint64_t)n1 * e;
        for (int a = 0; a < n0; ++a) o[a] = gi[a];
        for (int m = 0; m < n0; ++m) {
            uint64_t a = gi[m], b = gi[(m + 1) % n0];
            uint64_t key[2] = {a < b ? a : b, a < b ? b : a};
            uint64_t h = hash_parents(key, 2) & (cap - 1);
            for (;;) {
                parent_slot* sl = &table[h];
                if (!sl->used) {
                    sl->used = 1; sl->nkey = 2; sl->key[0] = key[0]; sl->key[1] = key[1]; sl->value = next;
                    for (int c = 0; c < 2; ++c) fv[2 * next + c] = (fv[2 * a + c] + fv[2 * b + c]) / 2.0;
                    o[n0 + m] = next++;
                    break;
                }
                if (sl->key[0] == key[0] && sl->key[1] == key[1]) { o[n0 + m] = sl->value; break; }
                h = (h + 1) & (cap - 1);
            }
        }
        if (from_kind == FO_QUAD4) {
            double ev[8], origin[2] = {0, 0};
            for (int n = 0; n < 4; ++n)
                for (int c = 0; c < 2; ++c) ev[2 * n + c] = vertices[2 * gi[n] + c];
            map_reference_coords(FO_QUAD4, ev, origin, fv + 2 * next);
            o[8] = next++;
        }
    }
    free(table);
    *out_vertices = fv; *out_num_vertices = next; *out_connectivity = conn;
    return FO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * materials  (fenris-solid/src/materials.rs, lib.rs)
 * ---------------------------------------------------------------------------------------------- */

/* materials.rs:31-43 */
void fo_lame_from_young_poisson(double young, double poisson, double* mu, double* lambda) {
    double m = 0.5 * young / (1.0 + poisson);
    double l = 2.0 * m * poisson / (1.0 - 2.0 * poisson);
    *mu = m; *lambda = l;
}

/* lib.rs:20-29 : F = I + u_grad^T */
static void deformation_gradient(int d, const double* u_grad, double* F) {
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) F[CM(i, j, d)] = (i == j ? 1.0 : 0.0) + u_grad[CM(j, i, d)];
}

/* materials.rs:71-79 : eps = F.symmetric_part() - I ; symmetric_part = (F^T + F) * 0.5 */
static void infinitesimal_strain(int d, const double* F, double* eps) {
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)
            eps[CM(i, j, d)] = (F[CM(j, i, d)] + F[CM(i, j, d)]) * 0.5 - (i == j ? 1.0 : 0.0);
}

/* materials.rs:383-390 : E = (F^T F - I) * 0.5 */
static void green_strain(int d, const double* F, double* E) {
    double Ft[9], FtF[9];
    transpose(d, F, Ft);
    matmul(d, d, d, Ft, F, FtF);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) E[CM(i, j, d)] = (FtF[CM(i, j, d)] - (i == j ? 1.0 : 0.0)) * 0.5;
}

static double mat_dot(int d, const double* A, const double* B) {
    /* nalgebra dotx on a d x d matrix: per column the 8-wide accumulators stay zero (nrows < 8), the
     * remainder loop adds entries one by one => plain left-to-right column-major sum from 0. */
    int n = d * d;
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += A[i] * B[i];
    return res;
}

/* fenris-solid/src/logdet.rs:17-86 ; returns 0 and leaves *out when undefined */
static int log_det_F(int d, const double* U, double* out) {
    if (d == 2) {
        double u11 = U[CM(0, 0, 2)], u22 = U[CM(1, 1, 2)], b = U[CM(0, 1, 2)], c = U[CM(1, 0, 2)];
        double gamma = u11 * u22 + u11 + u22 - b * c;
        if (gamma > -1.0) { *out = log1p(gamma); return 1; }
        return 0;
    }
    double u11 = U[CM(0, 0, 3)], u22 = U[CM(1, 1, 3)], u33 = U[CM(2, 2, 3)];
    double a = 1.0 + u11, e = 1.0 + u22, i = 1.0 + u33;
    double b = U[CM(0, 1, 3)], c = U[CM(0, 2, 3)], dd = U[CM(1, 0, 3)], f = U[CM(1, 2, 3)], g = U[CM(2, 0, 3)],
           h = U[CM(2, 1, 3)];
    double gamma = u11 * u22 * u33 + u11 * u22 + u11 * u33 + u22 * u33 + u11 + u22 + u33 + b * f * g + c * dd * h
                   - c * e * g - b * dd * i - a * f * h;
    if (gamma > -1.0) { *out = log1p(gamma); return 1; }
    return 0;
}

double fo_material_energy_density(int op, int d, const double* F, double mu, double lambda) {
    switch (op) {
        case FO_LINEAR_ELASTIC: { /* materials.rs:91-95 */
            double eps[9];
            infinitesimal_strain(d, F, eps);
            double tr = trace(d, eps);
            return mu * mat_dot(d, eps, eps) + 0.5 * lambda * (tr * tr);
        }
        case FO_NEO_HOOKEAN: { /* materials.rs:244-262 ; u_grad = (F - I)^T, du_dX = u_grad^T */
            double du_dX[9];
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) du_dX[CM(i, j, d)] = F[CM(i, j, d)] - (i == j ? 1.0 : 0.0);
            double logJ;
            if (log_det_F(d, du_dX, &logJ)) {
                double tr_E = trace(d, du_dX) + 0.5 * mat_dot(d, du_dX, du_dX);
                return mu * tr_E - mu * logJ + (0.5 * lambda) * (logJ * logJ);
            }
            return INFINITY;
        }
        case FO_STVK: { /* materials.rs:400-404 */
            double E[9];
            green_strain(d, F, E);
            double tr = trace(d, E);
            return mu * mat_dot(d, E, E) + 0.5 * lambda * (tr * tr);
        }
        default: return NAN;
    }
}

void fo_material_stress_tensor(int op, int d, const double* F, double mu, double lambda, double* P) {
    int dd = d * d;
    switch (op) {
        case FO_LINEAR_ELASTIC: { /* materials.rs:97-106 : eps*2*mu + diag(lambda*tr) */
            double eps[9];
            infinitesimal_strain(d, F, eps);
            double eps_tr = trace(d, eps);
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j)
                    P[CM(i, j, d)] = eps[CM(i, j, d)] * 2.0 * mu + (i == j ? lambda * eps_tr : 0.0);
            return;
        }
        case FO_NEO_HOOKEAN: { /* materials.rs:264-285 */
            double J = det(d, F);
            if (J <= 0.0) { for (int i = 0; i < dd; ++i) P[i] = NAN; return; }
            double logJ = log(J);
            double Finv[9], FinvT[9];
            try_inverse(d, F, Finv);
            transpose(d, Finv, FinvT);
            for (int i = 0; i < dd; ++i) P[i] = FinvT[i] * (-mu + lambda * logJ) + F[i] * mu;
            return;
        }
        case FO_STVK: { /* materials.rs:406-415 : F*E*2*mu + F*lambda*tr(E) */
            double E[9], FE[9];
            green_strain(d, F, E);
            matmul(d, d, d, F, E, FE);
            double tr = trace(d, E);
            for (int i = 0; i < dd; ++i) P[i] = FE[i] * 2.0 * mu + F[i] * lambda * tr;
            return;
        }
        default: for (int i = 0; i < dd; ++i) P[i] = NAN;
    }
}

void fo_material_stress_contraction(int op, int d, const double* F, const double* a, const double* b, double mu,
                                    double lambda, double* C) {
    int dd = d * d;
    switch (op) {
        case FO_LINEAR_ELASTIC: { /* materials.rs:108-118 : (I*(a.b) + b a^T)*mu + a b^T * lambda */
            double ab = dotd(d, a, b);
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j)
                    C[CM(i, j, d)] = ((i == j ? ab : 0.0 * ab) + b[i] * a[j]) * mu + a[i] * b[j] * lambda;
            return;
        }
        case FO_NEO_HOOKEAN: { /* materials.rs:287-315 */
            double J = det(d, F);
            if (J <= 0.0) { for (int i = 0; i < dd; ++i) C[i] = NAN; return; }
            double logJ = log(J);
            double Finv[9], FinvT[9], Fa[3], Fb[3];
            try_inverse(d, F, Finv);
            transpose(d, Finv, FinvT);
            matvec(d, FinvT, a, Fa);
            matvec(d, FinvT, b, Fb);
            double alpha = -mu + lambda * logJ;
            double mab = mu * dotd(d, a, b);
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j)
                    C[CM(i, j, d)] = Fa[i] * (Fb[j] * lambda) - Fb[i] * (Fa[j] * alpha) + (i == j ? mab : 0.0 * mab);
            return;
        }
        case FO_STVK: { /* materials.rs:417-438 */
            double E[9], Fa[3], Fb[3], Eb[3], Ft[9], FFt[9];
            green_strain(d, F, E);
            double a_dot_b = dotd(d, a, b);
            matvec(d, F, a, Fa);
            matvec(d, F, b, Fb);
            matvec(d, E, b, Eb);
            transpose(d, F, Ft);
            matmul(d, d, d, F, Ft, FFt);
            double c1 = 2.0 * mu * dotd(d, a, Eb) + lambda * trace(d, E) * a_dot_b;
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j)
                    C[CM(i, j, d)] = (i == j ? c1 : 0.0 * c1) + Fb[i] * Fa[j] * mu + Fa[i] * Fb[j] * lambda
                                     + FFt[CM(i, j, d)] * mu * a_dot_b;
            return;
        }
        default: for (int i = 0; i < dd; ++i) C[i] = NAN;
    }
}

/* ------------------------------------------------------------------------------------------------
 * local (element) assembly
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    int n, d, s;
    double ev[MAXN * MAXD]; /* gathered element vertices */
    double ue[MAXN * MAXD]; /* gathered element dofs */
    double phi_grad[MAXD * MAXN];
} elem_ws;

/* space_impl.rs:40-52,81-86 : gather element vertices from the mesh */
/* populate_element_data (quadrature_table.rs:283-291 uniform, :396-409 compact): the per-point data of element e */
static const double* element_params(const fo_assembler* a, uint64_t e, uint32_t q) {
    static const double zero_params[2] = {0, 0};
    if (a->elem_to_rule && a->rule_params) return a->rule_params + ((size_t)a->elem_to_rule[e] * a->nq + q) * 2;
    return a->q_params ? a->q_params + 2 * (size_t)q : zero_params;
}

static void gather_element(const fo_assembler* a, uint64_t e, int n, int d, double* ev) {
    const uint64_t* c = a->connectivity + (size_t)n * e;
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < d; ++i) ev[k * d + i] = a->vertices[(size_t)d * c[k] + (size_t)i];
}

/* global.rs:742-768 gather_global_to_local */
static void gather_u(const fo_assembler* a, uint64_t e, int n, int s, double* ue) {
    const uint64_t* c = a->connectivity + (size_t)n * e;
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < s; ++i) ue[k * s + i] = a->u ? a->u[(size_t)s * c[k] + (size_t)i] : 0.0;
}

/* compute_volume_u_grad elliptic.rs:25-59 : J^{-T} * sum_I grad_ref_I u_I^T  (d x s) */
static void compute_volume_u_grad(int d, int s, int n, const double* jinv_t, const double* gref, const double* ue,
                                  double* u_grad) {
    double acc[9];
    for (int i = 0; i < d * s; ++i) acc[i] = 0.0;
    for (int I = 0; I < n; ++I) /* ger(1, g_I, u_I, 1): acc[i][k] = 1*g[i]*u[k] + 1*acc */
        for (int k = 0; k < s; ++k)
            for (int i = 0; i < d; ++i) acc[CM(i, k, d)] = gref[CM(i, I, d)] * ue[I * s + k] + acc[CM(i, k, d)];
    matmul(d, d, s, jinv_t, acc, u_grad);
}

/* per-quadrature-point prologue shared by matrix / vector / energy (elliptic.rs:398-412) */
static int qp_prologue(const fo_assembler* a, uint64_t e, elem_ws* ws, const double* xi, double* j_det,
                       double* jinv_t, double* u_grad) {
    int d = ws->d, n = ws->n, s = ws->s;
    double J[9], Jinv[9];
    /* element.reference_jacobian(point): re-gathers the element (space_impl.rs:115-128) */
    gather_element(a, e, n, d, ws->ev);
    fo_element_reference_jacobian(a->elem_kind, ws->ev, xi, J);
    *j_det = det(d, J);
    if (!try_inverse(d, J, Jinv)) return FO_SINGULAR_JACOBIAN;
    transpose(d, Jinv, jinv_t);
    /* populate_basis_gradients: re-gathers again (space_impl.rs:95-113) and evaluates gradients */
    gather_element(a, e, n, d, ws->ev);
    fo_element_gradients(a->elem_kind, xi, ws->phi_grad);
    compute_volume_u_grad(d, s, n, jinv_t, ws->phi_grad, ws->ue, u_grad);
    return FO_OK;
}

static int ws_init(const fo_assembler* a, elem_ws* ws) {
    ws->n = fo_element_num_nodes(a->elem_kind);
    ws->d = fo_element_dim(a->elem_kind);
    if (ws->n < 0) return FO_BAD_ARGUMENT;
    ws->s = fo_operator_solution_dim(a->op_kind, ws->d);
    return FO_OK;
}

static void contraction(const fo_assembler* a, int d, const double* u_grad, const double* ga, const double* gb,
                        const double* params, double* C) {
    if (a->op_kind == FO_LAPLACE) { /* laplace.rs:60-68 */
        C[0] = dotd(d, ga, gb);
        return;
    }
    /* MaterialEllipticOperator::contract -> compute_stress_contraction_du -> F = I + grad u^T
     * (fenris-solid/src/lib.rs:127-135, 473-508): recomputed for every (I, J) pair */
    double F[9];
    deformation_gradient(d, u_grad, F);
    fo_material_stress_contraction(a->op_kind, d, F, ga, gb, params[0], params[1], C);
}

/* assemble_element_mass_matrix, src/assembly/local/mass.rs:204-286:
 * M_IJ = I_s * sum_q w |det J| rho phi_I phi_J, upper triangle then clone_upper_to_lower */
static int assemble_element_mass_matrix(const fo_assembler* a, uint64_t e, double* me) {
    elem_ws ws;
    if (ws_init(a, &ws)) return FO_BAD_ARGUMENT;
    int d = ws.d, n = ws.n, s = ws.s, ld = s * n;
    if (!a->q_params && !a->rule_params) return FO_BAD_ARGUMENT;
    for (int i = 0; i < ld * ld; ++i) me[i] = 0.0;
    double phi[MAXN], J[9];
    for (uint32_t q = 0; q < a->nq; ++q) {
        const double* xi = a->q_points + (size_t)d * q;
        gather_element(a, e, n, d, ws.ev);
        fo_element_reference_jacobian(a->elem_kind, ws.ev, xi, J);
        double j_det = det(d, J);
        fo_element_basis(a->elem_kind, xi, phi);
        double scale = a->q_weights[q] * fabs(j_det) * element_params(a, e, q)[0];
        for (int I = 0; I < n; ++I)
            for (int Jn = I; Jn < n; ++Jn) {
                double m = scale * phi[I] * phi[Jn];
                for (int i = 0; i < s; ++i) me[CM(s * I + i, s * Jn + i, ld)] += m;
            }
    }
    for (int j = 0; j < ld; ++j)
        for (int i = j + 1; i < ld; ++i) me[CM(i, j, ld)] = me[CM(j, i, ld)];
    return FO_OK;
}

/* assemble_element_elliptic_matrix, src/assembly/local/elliptic.rs:361-439 */
int fo_assemble_element_matrix(const fo_assembler* a, uint64_t e, double* ke) {
    if (a->op_kind == FO_MASS_SCALAR || a->op_kind == FO_MASS_VECTOR) return assemble_element_mass_matrix(a, e, ke);
    elem_ws ws;
    if (ws_init(a, &ws)) return FO_BAD_ARGUMENT;
    int d = ws.d, n = ws.n, s = ws.s, ld = s * n;
    /* assemble_element_matrix_into prologue (elliptic.rs:308-325) */
    gather_element(a, e, n, d, ws.ev);
    gather_u(a, e, n, s, ws.ue);
    for (int i = 0; i < ld * ld; ++i) ke[i] = 0.0; /* output.fill(0) :393 */
    for (uint32_t q = 0; q < a->nq; ++q) {
        double weight = a->q_weights[q];
        const double* xi = a->q_points + (size_t)d * q;
        const double* params = element_params(a, e, q);
        double j_det, jinv_t[9], u_grad[9];
        int st = qp_prologue(a, e, &ws, xi, &j_det, jinv_t, u_grad);
        if (st) return st;
        /* physical gradients, column by column (:415-418) */
        for (int I = 0; I < n; ++I) {
            double t[3];
            matvec(d, jinv_t, ws.phi_grad + (size_t)d * I, t);
            for (int i = 0; i < d; ++i) ws.phi_grad[CM(i, I, d)] = t[i];
        }
        double scale = weight * fabs(j_det); /* :422 */
        /* accumulate_contractions_into: operators.rs:176-188 / fenris-solid lib.rs:381-391.
         * All shipped operators are Symmetric => I in 0..=J; the data-defined operator may say NonSymmetric => I in 0..M (:178-181) */
        const int full = a->op_kind == FO_TENSOR && !a->tensor_symmetric;
        for (int Jn = 0; Jn < n; ++Jn)
            for (int In = 0; In <= (full ? n - 1 : Jn); ++In) {
                double C[9];
                if (a->op_kind == FO_TENSOR) {
                    /* C[i][k] = sum_jl a_I[j] A[i][j][k][l] b_J[l]: what `contract` of a gradient-independent operator returns */
                    const double* A = a->q_tensor + (size_t)q * d * d * d * d;
                    const double* ga = ws.phi_grad + (size_t)d * In;
                    const double* gb = ws.phi_grad + (size_t)d * Jn;
                    for (int i = 0; i < d; ++i)
                        for (int k = 0; k < d; ++k) {
                            double t = 0.0;
                            for (int j = 0; j < d; ++j)
                                for (int l = 0; l < d; ++l) t += ga[j] * A[((i * d + j) * d + k) * d + l] * gb[l];
                            C[CM(i, k, d)] = t;
                        }
                } else
                contraction(a, d, u_grad, ws.phi_grad + (size_t)d * In, ws.phi_grad + (size_t)d * Jn, params, C);
                if (a->op_kind == FO_LAPLACE) {
                    /* c_IJ += contraction * alpha */
                    ke[CM(In, Jn, ld)] += C[0] * scale;
                } else {
                    /* *c += alpha * y */
                    for (int j = 0; j < s; ++j)
                        for (int i = 0; i < s; ++i) ke[CM(s * In + i, s * Jn + j, ld)] += scale * C[CM(i, j, s)];
                }
            }
    }
    /* clone_upper_to_lower util.rs:38-51 -- `if matches!(operator.symmetry(), Symmetry::Symmetric)` (elliptic.rs:434-436) */
    if (!(a->op_kind == FO_TENSOR && !a->tensor_symmetric))
        for (int j = 0; j < ld; ++j)
            for (int i = j + 1; i < ld; ++i) ke[CM(i, j, ld)] = ke[CM(j, i, ld)];
    return FO_OK;
}

/* g^T = P for materials (lib.rs:462-470), grad u^T for Laplace (laplace.rs:52-56 + default transpose) */
static void elliptic_operator_transpose(const fo_assembler* a, int d, int s, const double* u_grad,
                                        const double* params, double* g_t /* s x d */) {
    if (a->op_kind == FO_LAPLACE) {
        for (int i = 0; i < d; ++i) g_t[CM(0, i, 1)] = u_grad[i];
        return;
    }
    (void)s;
    double F[9];
    deformation_gradient(d, u_grad, F);
    fo_material_stress_tensor(a->op_kind, d, F, params[0], params[1], g_t);
}

/* assemble_element_elliptic_vector, elliptic.rs:457-531 */
int fo_assemble_element_vector(const fo_assembler* a, uint64_t e, double* fe) {
    if (a->op_kind > FO_STVK) return FO_BAD_ARGUMENT;
    elem_ws ws;
    if (ws_init(a, &ws)) return FO_BAD_ARGUMENT;
    int d = ws.d, n = ws.n, s = ws.s;
    gather_element(a, e, n, d, ws.ev);
    gather_u(a, e, n, s, ws.ue);
    for (int i = 0; i < s * n; ++i) fe[i] = 0.0;
    for (uint32_t q = 0; q < a->nq; ++q) {
        double weight = a->q_weights[q];
        const double* xi = a->q_points + (size_t)d * q;
        const double* params = element_params(a, e, q);
        double j_det, jinv_t[9], u_grad[9], g_t[9], gj[9];
        int st = qp_prologue(a, e, &ws, xi, &j_det, jinv_t, u_grad);
        if (st) return st;
        elliptic_operator_transpose(a, d, s, u_grad, params, g_t);
        matmul(s, d, d, g_t, jinv_t, gj); /* g_t * j_inv_t : s x d */
        double alpha = weight * fabs(j_det);
        /* output.gemm(alpha, gj, phi_grad_ref, 1): per column j, axcpy chain over k:
         * y_i = (alpha * gj[i][k]) * B[k][j] + y_i */
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < d; ++k)
                for (int i = 0; i < s; ++i)
                    fe[CM(i, j, s)] = alpha * gj[CM(i, k, s)] * ws.phi_grad[CM(k, j, d)] + fe[CM(i, j, s)];
    }
    return FO_OK;
}

/* compute_element_elliptic_energy, elliptic.rs:551-605 */
int fo_assemble_element_scalar(const fo_assembler* a, uint64_t e, double* energy) {
    if (a->op_kind > FO_STVK) return FO_BAD_ARGUMENT;
    elem_ws ws;
    if (ws_init(a, &ws)) return FO_BAD_ARGUMENT;
    int d = ws.d, n = ws.n, s = ws.s;
    gather_element(a, e, n, d, ws.ev);
    gather_u(a, e, n, s, ws.ue);
    double integral = 0.0;
    for (uint32_t q = 0; q < a->nq; ++q) {
        double weight = a->q_weights[q];
        const double* xi = a->q_points + (size_t)d * q;
        const double* params = element_params(a, e, q);
        double j_det, jinv_t[9], u_grad[9], psi;
        int st = qp_prologue(a, e, &ws, xi, &j_det, jinv_t, u_grad);
        if (st) return st;
        if (a->op_kind == FO_LAPLACE) {
            psi = 0.5 * dotd(d, u_grad, u_grad); /* laplace.rs:35-37 */
        } else if (a->op_kind == FO_NEO_HOOKEAN) {
            /* compute_energy_density_du is overridden (materials.rs:249-262): du_dX = u_grad^T */
            double du_dX[9], logJ;
            transpose(d, u_grad, du_dX);
            if (log_det_F(d, du_dX, &logJ)) {
                double tr_E = trace(d, du_dX) + 0.5 * mat_dot(d, du_dX, du_dX);
                psi = params[0] * tr_E - params[0] * logJ + (0.5 * params[1]) * (logJ * logJ);
            } else psi = INFINITY;
        } else {
            double F[9];
            deformation_gradient(d, u_grad, F);
            psi = fo_material_energy_density(a->op_kind, d, F, params[0], params[1]);
        }
        integral += weight * fabs(j_det) * psi;
    }
    *energy = integral;
    return FO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * global assembly
 * ---------------------------------------------------------------------------------------------- */

/* CsrAssembler::assemble_pattern, global.rs:65-120.  The per-node FxHashSet is restated as
 * "collect, sort, unique" -- the set content (hence the output) is identical. */
int fo_assemble_pattern(uint64_t sdim, uint64_t num_nodes, uint64_t num_elements, const uint64_t* elem_offsets,
                        const uint64_t* elem_nodes, uint64_t* row_offsets, uint64_t* col_indices, uint64_t* nnz_out) {
    uint64_t N = num_nodes;
    uint64_t* cnt = calloc((size_t)N + 1, sizeof(uint64_t));
    if (!cnt) return FO_BAD_ARGUMENT;
    for (uint64_t e = 0; e < num_elements; ++e) {
        uint64_t b = elem_offsets[e], en = elem_offsets[e + 1];
        for (uint64_t i = b; i < en; ++i) {
            if (elem_nodes[i] >= N) { free(cnt); return FO_BAD_ARGUMENT; }
            cnt[elem_nodes[i] + 1] += en - b;
        }
    }
    for (uint64_t i = 0; i < N; ++i) cnt[i + 1] += cnt[i];
    uint64_t total = cnt[N];
    uint64_t* lists = malloc(sizeof(uint64_t) * (size_t)(total + 1));
    uint64_t* cursor = malloc(sizeof(uint64_t) * (size_t)(N + 1));
    uint64_t* ucount = malloc(sizeof(uint64_t) * (size_t)(N + 1));
    if (!lists || !cursor || !ucount) return FO_BAD_ARGUMENT;
    memcpy(cursor, cnt, sizeof(uint64_t) * (size_t)N);
    for (uint64_t e = 0; e < num_elements; ++e) {
        uint64_t b = elem_offsets[e], en = elem_offsets[e + 1];
        for (uint64_t i = b; i < en; ++i)
            for (uint64_t j = b; j < en; ++j) lists[cursor[elem_nodes[i]]++] = elem_nodes[j];
    }
    for (uint64_t i = 0; i < N; ++i) {
        uint64_t* l = lists + cnt[i];
        uint64_t m = cnt[i + 1] - cnt[i];
        qsort(l, (size_t)m, sizeof(uint64_t), cmp_u64);
        uint64_t u = 0;
        for (uint64_t k = 0; k < m; ++k)
            if (k == 0 || l[k] != l[k - 1]) l[u++] = l[k];
        ucount[i] = u;
    }
    uint64_t cur = 0, r = 0;
    row_offsets[0] = 0;
    for (uint64_t i = 0; i < N; ++i)
        for (uint64_t t = 0; t < sdim; ++t) {
            cur += sdim * ucount[i];
            row_offsets[++r] = cur;
        }
    if (nnz_out) *nnz_out = cur;
    if (col_indices) {
        uint64_t p = 0;
        for (uint64_t i = 0; i < N; ++i)
            for (uint64_t t = 0; t < sdim; ++t)
                for (uint64_t k = 0; k < ucount[i]; ++k)
                    for (uint64_t j = 0; j < sdim; ++j) col_indices[p++] = sdim * lists[cnt[i] + k] + j;
    }
    free(cnt); free(lists); free(cursor); free(ucount);
    return FO_OK;
}

/* sequential_greedy_coloring, fenris-paradis/src/coloring.rs:6-70 */
int fo_color_elements(uint64_t E, const uint64_t* elem_offsets, const uint64_t* elem_nodes, uint64_t* num_colors,
                      uint64_t* color_offsets, uint64_t* labels) {
    uint64_t max_node = 0;
    for (uint64_t i = 0; i < elem_offsets[E]; ++i)
        if (elem_nodes[i] > max_node) max_node = elem_nodes[i];
    int32_t* last = malloc(sizeof(int32_t) * (size_t)(max_node + 1));
    uint64_t* cur = malloc(sizeof(uint64_t) * (size_t)(E + 1));
    uint64_t* post = malloc(sizeof(uint64_t) * (size_t)(E + 1));
    if (!last || !cur || !post) return FO_BAD_ARGUMENT;
    for (uint64_t i = 0; i <= max_node; ++i) last[i] = -1;
    for (uint64_t i = 0; i < E; ++i) cur[i] = i;
    uint64_t ncur = E, nlab = 0, nc = 0;
    int32_t color_idx = 0;
    color_offsets[0] = 0;
    while (ncur > 0) {
        uint64_t npost = 0;
        for (uint64_t t = 0; t < ncur; ++t) {
            uint64_t e = cur[t];
            int blocked = 0;
            for (uint64_t i = elem_offsets[e]; i < elem_offsets[e + 1]; ++i)
                if (last[elem_nodes[i]] == color_idx) { blocked = 1; break; }
            if (blocked) post[npost++] = e;
            else {
                for (uint64_t i = elem_offsets[e]; i < elem_offsets[e + 1]; ++i) last[elem_nodes[i]] = color_idx;
                labels[nlab++] = e;
            }
        }
        color_offsets[++nc] = nlab;
        uint64_t* tmp = cur; cur = post; post = tmp;
        ncur = npost;
        ++color_idx;
    }
    *num_colors = nc;
    free(last); free(cur); free(post);
    return FO_OK;
}

/* add_element_row_to_csr_row, global.rs:504-537: forward iterator + linear find */
static int add_element_row_to_csr_row(double* row_values, const uint64_t* row_cols, uint64_t row_len,
                                      const uint64_t* nodes, const int* perm, int n, int dim, const double* ke, int ld,
                                      int local_row) {
    uint64_t it = 0;
    for (int p = 0; p < n; ++p) {
        int node_local = perm[p];
        uint64_t node_global = nodes[node_local];
        for (int i = 0; i < dim; ++i) {
            int local_col = dim * node_local + i;
            uint64_t global_col = (uint64_t)dim * node_global + (uint64_t)i;
            for (;;) {
                if (it >= row_len) return FO_COLUMN_NOT_FOUND; /* reference panics */
                uint64_t idx = it++;
                if (row_cols[idx] == global_col) {
                    row_values[idx] += ke[CM(local_row, local_col, ld)];
                    break;
                }
            }
        }
    }
    return FO_OK;
}

/* sort_unstable_by_key(|i| nodes[i]) -- keys are distinct for valid elements, so any sort agrees */
static void argsort_nodes(const uint64_t* nodes, int n, int* perm) {
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int i = 1; i < n; ++i) {
        int p = perm[i], j = i - 1;
        while (j >= 0 && nodes[perm[j]] > nodes[p]) { perm[j + 1] = perm[j]; --j; }
        perm[j + 1] = p;
    }
}

static int scatter_element(const fo_assembler* a, uint64_t e, int n, int s, const double* ke, const uint64_t* ro,
                           const uint64_t* ci, double* values) {
    const uint64_t* nodes = a->connectivity + (size_t)n * e;
    int perm[MAXN];
    argsort_nodes(nodes, n, perm);
    int ld = s * n;
    for (int ln = 0; ln < n; ++ln)
        for (int i = 0; i < s; ++i) {
            int local_row = s * ln + i;
            uint64_t grow = (uint64_t)s * nodes[ln] + (uint64_t)i;
            uint64_t b = ro[grow], en = ro[grow + 1];
            int st = add_element_row_to_csr_row(values + b, ci + b, en - b, nodes, perm, n, s, ke, ld, local_row);
            if (st) return st;
        }
    return FO_OK;
}

/* CsrAssembler::assemble_into_csr, global.rs:133-182 */
int fo_assemble_into_csr(const fo_assembler* a, const uint64_t* ro, const uint64_t* ci, double* values,
                         uint64_t* failed) {
    int n = fo_element_num_nodes(a->elem_kind), d = fo_element_dim(a->elem_kind);
    if (n < 0) return FO_BAD_ARGUMENT;
    int s = fo_operator_solution_dim(a->op_kind, d), ld = s * n;
    double* ke = malloc(sizeof(double) * (size_t)ld * (size_t)ld);
    int st = FO_OK;
    for (uint64_t e = 0; e < a->num_elements; ++e) {
        for (int i = 0; i < ld * ld; ++i) ke[i] = 0.0; /* resize_mut(.., zero) */
        st = fo_assemble_element_matrix(a, e, ke);
        if (!st) st = scatter_element(a, e, n, s, ke, ro, ci, values);
        if (st) { if (failed) *failed = e; break; }
    }
    free(ke);
    return st;
}

int fo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* CsrParAssembler::assemble_into_csr, global.rs:314-376: colours sequentially; inside a colour the
 * elements are distributed over threads (rayon work stealing ~ omp dynamic), thread-local K_e. */
int fo_par_assemble_into_csr(const fo_assembler* a, uint64_t num_colors, const uint64_t* color_offsets,
                             const uint64_t* labels, const uint64_t* ro, const uint64_t* ci, double* values,
                             int num_threads, uint64_t* failed) {
    int n = fo_element_num_nodes(a->elem_kind), d = fo_element_dim(a->elem_kind);
    if (n < 0) return FO_BAD_ARGUMENT;
    int s = fo_operator_solution_dim(a->op_kind, d), ld = s * n;
    int status = FO_OK;
    uint64_t fail_e = UINT64_MAX;
    if (num_threads <= 0) num_threads = fo_max_threads();
    for (uint64_t c = 0; c < num_colors && status == FO_OK; ++c) {
        int64_t b = (int64_t)color_offsets[c], en = (int64_t)color_offsets[c + 1];
#pragma omp parallel num_threads(num_threads)
        {
            double* ke = malloc(sizeof(double) * (size_t)ld * (size_t)ld);
#pragma omp for schedule(dynamic, 64)
            for (int64_t t = b; t < en; ++t) {
                uint64_t e = labels[t];
                for (int i = 0; i < ld * ld; ++i) ke[i] = 0.0;
                int st = fo_assemble_element_matrix(a, e, ke);
                if (!st) st = scatter_element(a, e, n, s, ke, ro, ci, values);
                if (st) {
#pragma omp critical
                    { if (e < fail_e) { fail_e = e; status = st; } }
                }
            }
            free(ke);
        }
    }
    if (status && failed) *failed = fail_e;
    return status;
}

/* assemble_element_source_vector, src/assembly/local/source.rs:219-278:
 *   f_e (s x n) = sum_q w |det J| f(x_q, data_q) phi(xi_q)   (output.gemm(w |det J|, f, phi, 1)).
 * The source function is arbitrary code in the reference; the closed family restated here:
 *   values == NULL:  f = density_q * g          (GravitySource, fenris-solid/src/gravity_source.rs:57-65)
 *   values != NULL:  f = values[(e nq + q) s ..] (source sampled at the physical points x_q) */
int fo_assemble_element_source_vector(const fo_assembler* a, uint64_t e, int s, const double* g, const double* values,
                                      double* fe) {
    int n = fo_element_num_nodes(a->elem_kind), d = fo_element_dim(a->elem_kind);
    if (n < 0 || s < 1 || s > MAXD) return FO_BAD_ARGUMENT;
    if (!values && (!g || (!a->q_params && !a->rule_params))) return FO_BAD_ARGUMENT;
    double ev[MAXN * MAXD], phi[MAXN], J[9], f[MAXD];
    gather_element(a, e, n, d, ev);
    for (int i = 0; i < s * n; ++i) fe[i] = 0.0; /* output.fill(0) :257 */
    for (uint32_t q = 0; q < a->nq; ++q) {
        const double* xi = a->q_points + (size_t)d * q;
        fo_element_basis(a->elem_kind, xi, phi);                       /* populate_basis :261 */
        fo_element_reference_jacobian(a->elem_kind, ev, xi, J);        /* :264 */
        for (int c = 0; c < s; ++c)
            f[c] = values ? values[((size_t)e * a->nq + q) * (size_t)s + (size_t)c] : g[c] * element_params(a, e, q)[0];
        double alpha = a->q_weights[q] * fabs(det(d, J));
        for (int I = 0; I < n; ++I)
            for (int c = 0; c < s; ++c) fe[s * I + c] += alpha * (f[c] * phi[I]);
    }
    return FO_OK;
}

/* VectorAssembler::assemble_vector_into (global.rs:582-608) driven by an ElementSourceAssembler */
int fo_assemble_source_vector_into(const fo_assembler* a, int s, const double* g, const double* values, double* out) {
    int n = fo_element_num_nodes(a->elem_kind);
    if (n < 0) return FO_BAD_ARGUMENT;
    double fe[MAXN * MAXD];
    for (uint64_t e = 0; e < a->num_elements; ++e) {
        int st = fo_assemble_element_source_vector(a, e, s, g, values, fe);
        if (st) return st;
        const uint64_t* nodes = a->connectivity + (size_t)n * e;
        for (int ln = 0; ln < n; ++ln)
            for (int i = 0; i < s; ++i) out[(size_t)s * nodes[ln] + (size_t)i] += fe[s * ln + i];
    }
    return FO_OK;
}

/* physical quadrature points x = map_reference_coords(xi_q) of every element (source.rs:263), E x nq x d */
int fo_physical_quadrature_points(const fo_assembler* a, double* x_out) {
    int n = fo_element_num_nodes(a->elem_kind), d = fo_element_dim(a->elem_kind);
    if (n < 0) return FO_BAD_ARGUMENT;
    double ev[MAXN * MAXD];
    for (uint64_t e = 0; e < a->num_elements; ++e) {
        gather_element(a, e, n, d, ev);
        for (uint32_t q = 0; q < a->nq; ++q)
            map_reference_coords(a->elem_kind, ev, a->q_points + (size_t)d * q, x_out + ((size_t)e * a->nq + q) * (size_t)d);
    }
    return FO_OK;
}

/* cuthill_mckee, src/mesh/reorder.rs:171-233, statement for statement (including the O(N) rescan for the next start
 * vertex).  sort_unstable_by_key leaves the order of equal-degree neighbours unspecified in the reference; a stable
 * insertion sort of the ascending column list is used here (ties by ascending index) -- it reproduces the known
 * answers of tests/unit_tests/reorder.rs.  perm[target] = source. */
int fo_cuthill_mckee(uint64_t n, const uint64_t* ro, const uint64_t* ci, uint64_t* perm) {
    unsigned char* visited = calloc(n + 1, 1);
    uint64_t* queue = malloc(sizeof(uint64_t) * (n + 1));
    uint64_t maxdeg = 0;
    for (uint64_t i = 0; i < n; ++i) if (ro[i + 1] - ro[i] > maxdeg) maxdeg = ro[i + 1] - ro[i];
    uint64_t* ws = malloc(sizeof(uint64_t) * (maxdeg + 1));
    uint64_t count = 0;
    for (;;) {
        /* least-degree unvisited vertex, the first one among equals (Iterator::min_by_key) :196-198 */
        uint64_t start = UINT64_MAX, best = UINT64_MAX;
        for (uint64_t v = 0; v < n; ++v)
            if (!visited[v] && ro[v + 1] - ro[v] < best) { best = ro[v + 1] - ro[v]; start = v; }
        if (start == UINT64_MAX) break;
        uint64_t head = 0, tail = 0;
        queue[tail++] = start;
        visited[start] = 1;
        while (head < tail) {
            uint64_t v = queue[head++];
            uint64_t m = ro[v + 1] - ro[v];
            for (uint64_t k = 0; k < m; ++k) ws[k] = ci[ro[v] + k];
            for (uint64_t a = 1; a < m; ++a) { /* stable insertion sort by degree */
                uint64_t x = ws[a], dx = ro[x + 1] - ro[x], b = a;
                while (b > 0 && ro[ws[b - 1] + 1] - ro[ws[b - 1]] > dx) { ws[b] = ws[b - 1]; --b; }
                ws[b] = x;
            }
            perm[count++] = v;
            for (uint64_t k = 0; k < m; ++k)
                if (!visited[ws[k]]) { visited[ws[k]] = 1; queue[tail++] = ws[k]; }
        }
    }
    free(visited); free(queue); free(ws);
    return count == n ? FO_OK : FO_BAD_ARGUMENT;
}

/* reorder_mesh_par, src/mesh/reorder.rs:54-95: RCM on the mesh's vertex graph (assemble_pattern with solution dim 1),
 * elements stably sorted by their smallest new vertex index. */
int fo_reorder_mesh(uint64_t N, uint64_t n, const uint64_t* conn, uint64_t E, uint64_t* vertex_perm, uint64_t* conn_perm) {
    uint64_t* offs = malloc(sizeof(uint64_t) * (E + 1));
    for (uint64_t e = 0; e <= E; ++e) offs[e] = e * n;
    uint64_t* ro = malloc(sizeof(uint64_t) * (N + 1));
    uint64_t nnz = 0;
    int st = fo_assemble_pattern(1, N, E, offs, conn, ro, NULL, &nnz);
    uint64_t* ci = malloc(sizeof(uint64_t) * (nnz + 1));
    if (!st) st = fo_assemble_pattern(1, N, E, offs, conn, ro, ci, &nnz);
    if (!st) st = fo_cuthill_mckee(N, ro, ci, vertex_perm);
    if (!st) {
        for (uint64_t a = 0, b = N; a + 1 < b; ++a) { --b; uint64_t t = vertex_perm[a]; vertex_perm[a] = vertex_perm[b]; vertex_perm[b] = t; }
        uint64_t* inv = malloc(sizeof(uint64_t) * (N + 1));
        for (uint64_t t = 0; t < N; ++t) inv[vertex_perm[t]] = t;
        uint64_t* key = malloc(sizeof(uint64_t) * (E + 1));
        for (uint64_t e = 0; e < E; ++e) {
            uint64_t m = UINT64_MAX;
            for (uint64_t a = 0; a < n; ++a) if (inv[conn[e * n + a]] < m) m = inv[conn[e * n + a]];
            key[e] = m;
            conn_perm[e] = e;
        }
        /* stable sort by key (sort_by_key :78-85): bottom-up merge sort */
        uint64_t* tmp = malloc(sizeof(uint64_t) * (E + 1));
        for (uint64_t w = 1; w < E; w *= 2) {
            for (uint64_t lo = 0; lo < E; lo += 2 * w) {
                uint64_t mid = lo + w < E ? lo + w : E, hi = lo + 2 * w < E ? lo + 2 * w : E, i = lo, j = mid, k = lo;
                while (i < mid && j < hi) tmp[k++] = (key[conn_perm[j]] < key[conn_perm[i]]) ? conn_perm[j++] : conn_perm[i++];
                while (i < mid) tmp[k++] = conn_perm[i++];
                while (j < hi) tmp[k++] = conn_perm[j++];
            }
            memcpy(conn_perm, tmp, sizeof(uint64_t) * E);
        }
        free(tmp); free(key); free(inv);
    }
    free(offs); free(ro); free(ci);
    return st;
}

/* ---- callers of the path: CG solve and error integrals -------------------------------------------------
 * ConjugateGradient::solve_with_guess, fenris-sparse/src/cg.rs:366-478, with operator = CSR matrix (spmm_csr_dense,
 * row by row, entries in column order), preconditioner = inverse diagonal (jacobi != 0; tests/convergence_tests/
 * poisson_mms_common.rs:148-151) or identity, RelativeResidualCriterion(tol) (cg.rs:86-124).
 * Returns 0 ok, 7 max iterations, 8 indefinite operator, 9 indefinite preconditioner (codes of fenris_hip.h). */
static void csr_spmv(uint64_t n, const uint64_t* ro, const uint64_t* ci, const double* v, const double* x, double* y) {
    for (uint64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (uint64_t k = ro[i]; k < ro[i + 1]; ++k) s += v[k] * x[ci[k]];
        y[i] = s;
    }
}
static double dotn(uint64_t n, const double* a, const double* b) {
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}
int fo_cg_solve(uint64_t n, const uint64_t* ro, const uint64_t* ci, const double* values, const double* b, double* x,
                int jacobi, double tol, uint64_t max_iter, uint64_t* num_iterations) {
    double* r = malloc(sizeof(double) * (n + 1)), *z = malloc(sizeof(double) * (n + 1));
    double* p = malloc(sizeof(double) * (n + 1)), *Ap = malloc(sizeof(double) * (n + 1)), *dinv = NULL;
    int status = 0;
    uint64_t it = 0;
    if (jacobi) {
        dinv = malloc(sizeof(double) * (n + 1));
        for (uint64_t i = 0; i < n; ++i) {
            dinv[i] = 0.0;
            for (uint64_t k = ro[i]; k < ro[i + 1]; ++k)
                if (ci[k] == i) dinv[i] = 1.0 / values[k];
        }
    }
    csr_spmv(n, ro, ci, values, x, r);                               /* r <- A x        :388 */
    for (uint64_t i = 0; i < n; ++i) r[i] = b[i] - r[i];             /* r <- b - r      :392 */
    for (uint64_t i = 0; i < n; ++i) z[i] = dinv ? dinv[i] * r[i] : r[i]; /* z = P r    :395 */
    memcpy(p, z, sizeof(double) * n);                                /* p = z           :400 */
    double zTr = dotn(n, z, r);
    double b_norm = sqrt(dotn(n, b, b));
    if (b_norm == 0.0) {                                             /* :409-412 */
        for (uint64_t i = 0; i < n; ++i) x[i] = 0.0;
        goto done;
    }
    for (;;) {
        if (sqrt(dotn(n, r, r)) <= tol * b_norm) break;              /* :108-124 */
        if (max_iter && it >= max_iter) { status = 7; break; }       /* :427-431 */
        csr_spmv(n, ro, ci, values, p, Ap);
        double pAp = dotn(n, p, Ap);
        if (pAp <= 0.0) { status = 8; break; }
        if (zTr <= 0.0) { status = 9; break; }
        double alpha = zTr / pAp;
        for (uint64_t i = 0; i < n; ++i) x[i] += alpha * p[i];
        for (uint64_t i = 0; i < n; ++i) r[i] -= alpha * Ap[i];
        ++it;
        for (uint64_t i = 0; i < n; ++i) z[i] = dinv ? dinv[i] * r[i] : r[i];
        double zTr_next = dotn(n, z, r);
        double beta = zTr_next / zTr;
        for (uint64_t i = 0; i < n; ++i) { p[i] *= beta; p[i] += z[i]; }
        zTr = zTr_next;
    }
done:
    if (num_iterations) *num_iterations = it;
    free(r); free(z); free(p); free(Ap); free(dinv);
    return status;
}

/* estimate_L2_error_squared / estimate_H1_seminorm_error_squared, src/error.rs:287-372 via assemble_scalar
 * (element order).  which = 0: sum_q w |det J| |u_h - u|^2 (u_h = sum_n phi_n u_n);  which = 1: sum_q w |det J|
 * |grad u_h - grad u|_F^2 with grad u_h = sum_n (J^-T grad phi_n) u_n^T (d x s).  The reference solution is given
 * sampled at the physical points: (E, nq, s) values or (E, nq, d, s) gradients. */
int fo_estimate_error_squared(const fo_assembler* a, int which, int s, const double* uh, const double* exact, double* out) {
    int n = fo_element_num_nodes(a->elem_kind), d = fo_element_dim(a->elem_kind);
    if (n < 0 || s < 1 || s > MAXD) return FO_BAD_ARGUMENT;
    double total = 0.0;
    double ev[MAXN * MAXD], phi[MAXN], G[MAXD * MAXN], J[9], Jinv[9];
    for (uint64_t e = 0; e < a->num_elements; ++e) {
        gather_element(a, e, n, d, ev);
        const uint64_t* nodes = a->connectivity + (size_t)n * e;
        double elem = 0.0;
        for (uint32_t q = 0; q < a->nq; ++q) {
            const double* xi = a->q_points + (size_t)d * q;
            fo_element_reference_jacobian(a->elem_kind, ev, xi, J);
            double j_det = det(d, J);
            double err2 = 0.0;
            size_t base = (size_t)e * a->nq + q;
            if (which == 0) {
                fo_element_basis(a->elem_kind, xi, phi);
                for (int k = 0; k < s; ++k) {
                    double u = 0.0;
                    for (int I = 0; I < n; ++I) u += phi[I] * uh[(size_t)s * nodes[I] + (size_t)k];
                    double dd = u - exact[base * (size_t)s + (size_t)k];
                    err2 += dd * dd;
                }
            } else {
                if (!try_inverse(d, J, Jinv)) return FO_SINGULAR_JACOBIAN;
                fo_element_gradients(a->elem_kind, xi, G);
                for (int i = 0; i < d; ++i)
                    for (int k = 0; k < s; ++k) {
                        double gu = 0.0;
                        for (int I = 0; I < n; ++I) {
                            double g = 0.0; /* (J^-T grad_ref)_i = sum_c Jinv[c][i] G[c][I] */
                            for (int c = 0; c < d; ++c) g += Jinv[CM(c, i, d)] * G[CM(c, I, d)];
                            gu += g * uh[(size_t)s * nodes[I] + (size_t)k];
                        }
                        double dd = gu - exact[(base * (size_t)d + (size_t)i) * (size_t)s + (size_t)k];
                        err2 += dd * dd;
                    }
            }
            elem += a->q_weights[q] * fabs(j_det) * err2;
        }
        total += elem;
    }
    *out = total;
    return FO_OK;
}

/* VectorAssembler::assemble_vector_into, global.rs:582-608 + add_local_to_global :770-796 */
int fo_assemble_vector_into(const fo_assembler* a, double* out, uint64_t* failed) {
    int n = fo_element_num_nodes(a->elem_kind), d = fo_element_dim(a->elem_kind);
    if (n < 0) return FO_BAD_ARGUMENT;
    int s = fo_operator_solution_dim(a->op_kind, d);
    double fe[MAXN * MAXD];
    for (uint64_t e = 0; e < a->num_elements; ++e) {
        int st = fo_assemble_element_vector(a, e, fe);
        if (st) { if (failed) *failed = e; return st; }
        const uint64_t* nodes = a->connectivity + (size_t)n * e;
        for (int ln = 0; ln < n; ++ln)
            for (int i = 0; i < s; ++i) out[(size_t)s * nodes[ln] + (size_t)i] += fe[s * ln + i];
    }
    return FO_OK;
}

/* VectorParAssembler::assemble_vector_into, global.rs:643-685 */
int fo_par_assemble_vector_into(const fo_assembler* a, uint64_t num_colors, const uint64_t* color_offsets,
                                const uint64_t* labels, double* out, int num_threads, uint64_t* failed) {
    int n = fo_element_num_nodes(a->elem_kind), d = fo_element_dim(a->elem_kind);
    if (n < 0) return FO_BAD_ARGUMENT;
    int s = fo_operator_solution_dim(a->op_kind, d);
    int status = FO_OK;
    uint64_t fail_e = UINT64_MAX;
    if (num_threads <= 0) num_threads = fo_max_threads();
    for (uint64_t c = 0; c < num_colors && status == FO_OK; ++c) {
        int64_t b = (int64_t)color_offsets[c], en = (int64_t)color_offsets[c + 1];
#pragma omp parallel for schedule(dynamic, 64) num_threads(num_threads)
        for (int64_t t = b; t < en; ++t) {
            uint64_t e = labels[t];
            double fe[MAXN * MAXD];
            int st = fo_assemble_element_vector(a, e, fe);
            if (st) {
#pragma omp critical
                { if (e < fail_e) { fail_e = e; status = st; } }
                continue;
            }
            const uint64_t* nodes = a->connectivity + (size_t)n * e;
            for (int ln = 0; ln < n; ++ln)
                for (int i = 0; i < s; ++i) out[(size_t)s * nodes[ln] + (size_t)i] += fe[s * ln + i];
        }
    }
    if (status && failed) *failed = fail_e;
    return status;
}

/* assemble_scalar, global.rs:697-711 */
int fo_assemble_scalar(const fo_assembler* a, double* out, uint64_t* failed) {
    double total = 0.0;
    for (uint64_t e = 0; e < a->num_elements; ++e) {
        double c;
        int st = fo_assemble_element_scalar(a, e, &c);
        if (st) { if (failed) *failed = e; return st; }
        total += c;
    }
    *out = total;
    return FO_OK;
}

/* apply_homogeneous_dirichlet_bc_csr, global.rs:379-451 */
int fo_apply_homogeneous_dirichlet_bc_csr(uint64_t num_rows, const uint64_t* ro, const uint64_t* ci, double* values,
                                          const uint64_t* nodes, uint64_t nbc, uint64_t d) {
    double scale = 1.0;
    int found = 0;
    for (uint64_t r = 0; r < num_rows && !found; ++r)
        for (uint64_t k = ro[r]; k < ro[r + 1]; ++k)
            if (ci[k] == r) {
                if (values[k] != 0.0) { scale = fabs(values[k]); found = 1; }
                break;
            }
    uint8_t* member = calloc((size_t)num_rows + 1, 1);
    uint8_t* visit = calloc((size_t)num_rows + 1, 1);
    if (!member || !visit) return FO_BAD_ARGUMENT;
    for (uint64_t t = 0; t < nbc; ++t)
        for (uint64_t i = 0; i < d; ++i) {
            uint64_t row = d * nodes[t] + i;
            member[row] = 1;
            for (uint64_t k = ro[row]; k < ro[row + 1]; ++k) {
                if (ci[k] == row) values[k] = scale;
                else { values[k] = 0.0; visit[ci[k]] = 1; }
            }
        }
    for (uint64_t r = 0; r < num_rows; ++r)
        if (visit[r] && !member[r])
            for (uint64_t k = ro[r]; k < ro[r + 1]; ++k)
                if (member[ci[k]]) values[k] = 0.0;
    free(member); free(visit);
    return FO_OK;
}
