// Register-resident element pass for the residual vector, the energy and the source vector of the small iso-parametric
// elements (Quad4, Tri3, Tet4, Hex8): ONE THREAD PER ELEMENT.  Replaces, for these elements, the LDS-staged kernels
// k_assemble_vector_stream / k_assemble_scalar / k_assemble_source of assemble_kernels.hpp (round 2: residual 3.1 ms, energy
// 3.4 ms, gravity source 5.5 ms on Hex8 216^3 -- all of them bound by instruction issue and LDS round trips, none by memory).
//
//  * a thread fetches its element's node indices, vertex coordinates and u once (8-byte gathers whose lanes walk consecutive
//    nodes of consecutive elements), keeps them in registers and walks the quadrature points: J = X G^T, its inverse,
//    grad u = J^-T (sum_n ghat_n u_n^T), the operator's stress / energy density, and f_n += (s P J^-T) ghat_n -- what
//    assemble_element_elliptic_vector / compute_element_elliptic_energy do per point (src/assembly/local/elliptic.rs:457-605),
//    with the physical gradients never formed.  The reference-gradient table is uniform over the wavefront: it comes through
//    scalar loads and enters the multiplications as a scalar operand.
//  * vectors, two passes without atomics, both coalesced: the element vectors go to a scratch array laid out BY LOCAL NODE,
//    fe[a][e][c] -- consecutive threads (elements) write consecutive 8 S bytes -- and k_vector_from_elements_soa gives every
//    node one thread that sums its (element, local node) entries in ascending element order (the reference's sequential
//    order, bitwise reproducible): for neighbouring nodes of a structured mesh the k-th entries are neighbouring elements with
//    the same local node, i.e. neighbouring places of fe[a].  (Round 2 stored fe[e][a][c]: 24-byte pieces 192 bytes apart on
//    both sides.)
//  * energy: element energies summed per wavefront and workgroup in a fixed tree, workgroup partials in index order by a second
//    one-workgroup kernel: one double comes back to the host.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_common.hpp"
#include "small_ops.hpp"

namespace fenris_hip {

enum { EP_VECTOR = 0, EP_SCALAR = 1 };

// The quadrature tables (weights, reference gradients, basis values, uniform parameters) are read at wavefront-uniform addresses and
// never written while a kernel runs: through the constant address space they come by scalar loads (s_load) and enter the
// multiplications as scalar operands -- as plain global loads every table entry was a 64-lane vector load of one address.
typedef const __attribute__((address_space(4))) double* ep_table;
__device__ __forceinline__ ep_table ep_const(const double* p) { return (ep_table)p; }

// stress P (s x d) and energy density psi of one quadrature point from grad u (d x s):
// laplace.rs:26-73; fenris-solid/src/materials.rs:71-123 (LinearElastic), 236-353 (NeoHookean, J <= 0 => NaN block / inf),
// 392-469 (StVK).  Same formulas as phase B of assemble_kernels.hpp.
template <int OP, int D, int S, int WHAT>
__device__ __forceinline__ void material_point(const double (&gu)[D][S], double mu, double lambda, double (&P)[S][D], double& psi) {
    psi = 0.0;
    if constexpr (OP == FH_LAPLACE) {
#pragma unroll
        for (int k = 0; k < D; ++k) { P[0][k] = gu[k][0]; psi = fma(gu[k][0], gu[k][0], psi); }
        psi *= 0.5;
    } else {
        double F[D][D];  // F = I + (grad u)^T  (fenris-solid/src/lib.rs:20-29)
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) F[i][j] = (i == j ? 1.0 : 0.0) + gu[j][i];
        if constexpr (OP == FH_LINEAR_ELASTIC) {
            double eps[D][D];
            double tr = 0.0, ee = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    eps[i][j] = (F[j][i] + F[i][j]) * 0.5 - (i == j ? 1.0 : 0.0);
                    ee = fma(eps[i][j], eps[i][j], ee);
                }
#pragma unroll
            for (int i = 0; i < D; ++i) tr += eps[i][i];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) P[i][j] = eps[i][j] * 2.0 * mu + (i == j ? lambda * tr : 0.0);
            psi = mu * ee + 0.5 * lambda * (tr * tr);
        } else if constexpr (OP == FH_NEO_HOOKEAN) {
            const double Jd = det_small<D>(F);
            if (Jd <= 0.0) {
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) P[i][j] = __builtin_nan("");
            } else {
                double Fi[D][D];
                inv_small(F, Jd, Fi);
                const double c = -mu + lambda * log(Jd);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) P[i][j] = Fi[j][i] * c + F[i][j] * mu;
            }
            if constexpr (WHAT == EP_SCALAR) {
                // materials.rs:249-262 with log_det_F of du_dX = (grad u)^T (logdet.rs:17-86)
                double U[D][D];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) U[i][j] = gu[j][i];
                double gamma;
                if constexpr (D == 2) {
                    gamma = U[0][0] * U[1][1] + U[0][0] + U[1][1] - U[0][1] * U[1][0];
                } else {
                    const double u11 = U[0][0], u22 = U[1][1], u33 = U[2][2];
                    const double aa = 1.0 + u11, e2 = 1.0 + u22, i2 = 1.0 + u33;
                    const double b = U[0][1], c = U[0][2], d2 = U[1][0], f = U[1][2], g = U[2][0], h = U[2][1];
                    gamma = u11 * u22 * u33 + u11 * u22 + u11 * u33 + u22 * u33 + u11 + u22 + u33 + b * f * g + c * d2 * h -
                            c * e2 * g - b * d2 * i2 - aa * f * h;
                }
                if (gamma > -1.0) {
                    const double logJ = log1p(gamma);
                    double trU = 0.0, nn = 0.0;
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        trU += U[i][i];
#pragma unroll
                        for (int j = 0; j < D; ++j) nn = fma(U[i][j], U[i][j], nn);
                    }
                    psi = mu * (trU + 0.5 * nn) - mu * logJ + (0.5 * lambda) * (logJ * logJ);
                } else {
                    psi = __builtin_inf();
                }
            }
        } else {  // StVK: P = F E 2 mu + F lambda tr E ; psi = mu E:E + lambda/2 tr^2
            double Eg[D][D];
            double trE = 0.0, ee = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = fma(F[k][i], F[k][j], t);
                    Eg[i][j] = (t - (i == j ? 1.0 : 0.0)) * 0.5;
                    ee = fma(Eg[i][j], Eg[i][j], ee);
                }
#pragma unroll
            for (int i = 0; i < D; ++i) trE += Eg[i][i];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = fma(F[i][k], Eg[k][j], t);
                    P[i][j] = t * 2.0 * mu + F[i][j] * lambda * trE;
                }
            psi = mu * ee + 0.5 * lambda * (trE * trE);
        }
    }
}

// sum of v over the workgroup (256 threads) in a fixed tree: lanes by xor-shuffles, wavefronts in index order
__device__ __forceinline__ double block_sum_256(double v, double* lds4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

template <int EK, int OP, int WHAT>
struct EPDims {
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    static constexpr int D = E::D, N = E::N, S = O::S, NF = (WHAT == EP_VECTOR ? N : 1);
};

// vertex coordinates and u of element ec's nodes into the registers of the thread (8-byte gathers)
template <int EK, int OP, int WHAT>
__device__ __forceinline__ void element_pass_load(const KArgs& a, const long long ec, double (&X)[EPDims<EK, OP, WHAT>::N][EPDims<EK, OP, WHAT>::D],
                                                  double (&U)[EPDims<EK, OP, WHAT>::N][EPDims<EK, OP, WHAT>::S]) {
    constexpr int D = EPDims<EK, OP, WHAT>::D, N = EPDims<EK, OP, WHAT>::N, S = EPDims<EK, OP, WHAT>::S;
    int nd[N];
    if constexpr (N % 4 == 0) {   // a connectivity row of 16 or 32 bytes: one or two 16-byte loads
        const int4* row = reinterpret_cast<const int4*>(a.conn + (size_t)ec * N);
#pragma unroll
        for (int n = 0; n < N; n += 4) {
            const int4 v = row[n / 4];
            nd[n] = v.x; nd[n + 1] = v.y; nd[n + 2] = v.z; nd[n + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int n = 0; n < N; ++n) nd[n] = a.conn[(size_t)ec * N + n];
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
#pragma unroll
        for (int i = 0; i < D; ++i) X[n][i] = a.verts[(size_t)nd[n] * D + i];
#pragma unroll
        for (int k = 0; k < S; ++k) U[n][k] = a.u ? a.u[(size_t)nd[n] * S + k] : 0.0;
    }
}

// one element in the registers of one thread: f[a][c] (EP_VECTOR) or the element's energy (EP_SCALAR) from its vertex coordinates X and
// its u.  `ec` is the element whose data the thread holds (every lane computes: uniform control flow), `live` whether it is the
// thread's own (singular reports)
// (U: anything with  double operator()(n, k): the registers of the thread -- EPRegU -- or a staging place in LDS)
template <int N, int S>
struct EPRegU {
    const double (&u)[N][S];
    __device__ __forceinline__ double operator()(int n, int k) const { return u[n][k]; }
};

// ---- Hex8 in the monomial basis (round 5).  A trilinear field is  v = c0 + c1 xi + c2 eta + c3 zeta + c4 xi eta + c5 xi zeta + c6 eta zeta
// + c7 xi eta zeta  with c = (1/8) W v_nodes, W the sign matrix of the reference nodes (hexahedron.rs:49-58): a Walsh-Hadamard butterfly, 24
// additions per field and ELEMENT.  Its reference gradient at a point is then three chains of three FMAs,
//     d/dxi = c1 + c4 eta + c5 zeta + c7 eta zeta   (and cyclic),
// 9 FMAs per field and point instead of the 24 of  sum_n ghat_n(xi_q) v_n  (elliptic.rs:398-422 evaluates exactly these sums); the same
// identity read backwards turns  f_n += M_q ghat_n(xi_q)  (elliptic.rs:506-527) into twelve moment sums per component and point and one
// butterfly per component at the end.  Any quadrature rule (the points' coordinates and pair products come with the table: a.qmono).
// Same sums in another order: equal to the generic body to rounding, bitwise reproducible, checked against the oracle at 1e-12.
// index of a reference node in binary order (bit 0: xi > 0, bit 1: eta > 0, bit 2: zeta > 0)
__device__ __forceinline__ constexpr int hex8_bin(int n) { return n == 2 ? 3 : n == 3 ? 2 : n == 6 ? 7 : n == 7 ? 6 : n; }
// w[k] = sum_b (prod over the axes in k of the sign of node b along that axis) v[b], in place; k in binary order
__device__ __forceinline__ void hex8_wht(double (&v)[8]) {
#pragma unroll
    for (int ax = 1; ax < 8; ax <<= 1)
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (!(b & ax)) {
                const double lo = v[b], hi = v[b | ax];
                v[b] = hi + lo;        // the axis is not in k
                v[b | ax] = hi - lo;   // the axis is in k: sign +1 on the upper node, -1 on the lower
            }
}
// AFF: every element of the mesh is a parallelepiped (k_classify_affine_hex8 says so for all of them: KArgs::all_affine) -- J from the three
// linear coefficients of the map, once; no per-wavefront test, no second path (which alone keeps 24 more registers alive).
// AFFM = 2 (round 5, residual of Laplace / LinearElastic with one parameter pair for every point, all elements affine): NO loop over the
// quadrature points.  With J constant the integrand's matrix M(xi) = s P(J^-T R(xi)) J^-T is LINEAR in the reference gradient R(xi) of u, and
// R(xi) = R_1 + R_xi xi + R_eta eta + R_zeta zeta + R_ez eta zeta + R_xz xi zeta + R_xe xi eta  with coefficient matrices that are rows of the
// monomial coefficients of u (d/dxi = c1 + c3 eta + c5 zeta + c7 eta zeta, and cyclic).  The moment sums of the point loop then are
//     sum_q w_q M(xi_q) (monomial)  =  sum_t  L(R_t)  x  (moment of the rule: sum_q w_q monomial_t monomial),
// and for a rule that is symmetric in every coordinate (all moments with an odd power vanish: checked on the host, fh_ctx::qmom_ok) only the
// seven squares survive: seven applications of the linear map L to sparse matrices -- and of each result only the columns that meet a
// non-vanishing moment -- instead of eight full point evaluations: ~640 instead of ~1 200 vector instructions per element.  The same sums as
// elliptic.rs:506-527 in another order (like the affine stiffness kernel: equal to rounding, checked against the oracle at 1e-12).
template <int OP, int WHAT, int AFFM = 0, class UAcc>
__device__ __forceinline__ void element_pass_body_hex8(const KArgs& a, const long long e, const bool live, const long long ec,
                                                       const double (&X)[8][3], const UAcc& U,
                                                       double (&f)[EPDims<FH_HEX8, OP, WHAT>::NF][EPDims<FH_HEX8, OP, WHAT>::S], double& energy) {
    using O = OpT<OP, 3>;
    constexpr int D = 3, N = 8, S = O::S;
    constexpr bool AFF = AFFM != 0;
    constexpr bool POLY = AFFM == 2 && (OP == FH_LAPLACE || OP == FH_LINEAR_ELASTIC);   // (the energy too: a quadratic form, its cross moments vanish likewise)
    energy = 0.0;
    const double* par_e = a.rule_map ? a.rparams + (size_t)a.rule_map[ec] * a.nq * 2 : nullptr;
    // coefficients of the coordinate map and of u (c0 is not needed: only gradients enter)
    double cx[D][8], cu[S][8];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int n = 0; n < N; ++n) cx[i][hex8_bin(n)] = X[n][i];
        hex8_wht(cx[i]);
#pragma unroll
        for (int k = 1; k < 8; ++k) cx[i][k] *= 0.125;
        if constexpr (AFF) {   // (the mixed coefficients are rounding noise below 2^-46 of the edges: dropped, like in the affine stiffness kernel)
            cx[i][3] = 0.0; cx[i][5] = 0.0; cx[i][6] = 0.0; cx[i][7] = 0.0;
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < S; ++k2) {
#pragma unroll
        for (int n = 0; n < N; ++n) cu[k2][hex8_bin(n)] = U(n, k2);
        hex8_wht(cu[k2]);
#pragma unroll
        for (int k = 1; k < 8; ++k) cu[k2][k] *= 0.125;
    }
    // affine element (a parallelepiped: the mixed coefficients of the map vanish; the test of k_classify_affine_hex8, 2^-46 of the edges
    // from node 0): J is the same at every point.  Uniform over the wavefront, like in the generic body.
    double len = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) len += fabs(X[1][i] - X[0][i]) + fabs(X[3][i] - X[0][i]) + fabs(X[4][i] - X[0][i]);
    const double tol = len * 0x1p-46 * 0.125;   // (the coefficients carry the factor 1/8)
    bool aff = true;
#pragma unroll
    for (int i = 0; i < D; ++i) aff = aff && fabs(cx[i][3]) <= tol && fabs(cx[i][5]) <= tol && fabs(cx[i][6]) <= tol && fabs(cx[i][7]) <= tol;
    const bool const_j = AFF || __all(aff ? 1 : 0) != 0;
    // reference gradient of a field with coefficients c at the point (xi, eta, zeta; ez = eta zeta, xz = xi zeta, xe = xi eta)
    auto grad = [](const double (&c)[8], double xi, double eta, double zeta, double ez, double xz, double xe, double (&g)[3]) {
        g[0] = fma(c[7], ez, fma(c[5], zeta, fma(c[3], eta, c[1])));
        g[1] = fma(c[7], xz, fma(c[6], zeta, fma(c[3], xi, c[2])));
        g[2] = fma(c[7], xe, fma(c[6], eta, fma(c[5], xi, c[4])));
    };
    double J[D][D], Ji[D][D], adet = 0.0;
    // moments of M = s P J^-T against the monomials of the reference gradients, summed straight into the coefficients of the sign products:
    //   dk[i][1] (xi) += M_i0, [2] (eta) += M_i1, [4] (zeta) += M_i2, [3] (xi eta) += M_i0 eta + M_i1 xi, [5] (xi zeta) += M_i0 zeta + M_i2 xi,
    //   [6] (eta zeta) += M_i1 zeta + M_i2 eta, [7] += M_i0 eta zeta + M_i1 xi zeta + M_i2 xi eta       (21 accumulators; [0] stays zero)
    double dk[S][8];
    if constexpr (WHAT == EP_VECTOR) {
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int t = 0; t < 8; ++t) dk[i][t] = 0.0;
    }
    auto point = [&](int q, auto need_j_tag) {
        constexpr bool need_j = decltype(need_j_tag)::value;
        const ep_table Q = ep_const(a.qmono) + (size_t)q * 8;   // uniform over the wavefront: scalar loads
        const double xi = Q[0], eta = Q[1], zeta = Q[2], ez = Q[3], xz = Q[4], xe = Q[5];
        if constexpr (need_j) {
#pragma unroll
            for (int i = 0; i < D; ++i) grad(cx[i], xi, eta, zeta, ez, xz, xe, J[i]);   // J[i][j] = d x_i / d xi_j (hexahedron.rs:101-107)
            const double detJ = det_small<D>(J);
            if (detJ == 0.0) {  // try_inverse fails only for det == 0 exactly (elliptic.rs:401-404)
                if (live) report_singular(a.status, e);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) Ji[i][j] = 0.0;
            } else {
                inv_small(J, detJ, Ji);
            }
            adet = fabs(detJ);
        }
        double R[D][S];   // R[j][k] = d u_k / d xi_j = sum_n ghat_n u_n^T
#pragma unroll
        for (int k = 0; k < S; ++k) {
            double g[3];
            grad(cu[k], xi, eta, zeta, ez, xz, xe, g);
#pragma unroll
            for (int j = 0; j < D; ++j) R[j][k] = g[j];
        }
        const double s = ep_const(a.qw)[q] * adet;   // w |det J| (elliptic.rs:422)
        double gu[D][S];                             // grad u = J^-T R
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < S; ++k) {
                double t = 0.0;
#pragma unroll
                for (int m = 0; m < D; ++m) t = fma(Ji[m][i], R[m][k], t);
                gu[i][k] = t;
            }
        double mu = 0.0, lambda = 0.0;
        if (OP != FH_LAPLACE) {
            if (par_e) { mu = par_e[2 * q]; lambda = par_e[2 * q + 1]; }
            else { mu = ep_const(a.qparams)[2 * q]; lambda = ep_const(a.qparams)[2 * q + 1]; }
        }
        double P[S][D], psi;
        material_point<OP, D, S, WHAT>(gu, mu, lambda, P, psi);
        if constexpr (WHAT == EP_SCALAR) {
            energy = fma(s, psi, energy);
        } else {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                double Mi[D];   // (s P J^-T)[i][m]
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = fma(P[i][k], Ji[m][k], t);
                    Mi[m] = s * t;
                }
                dk[i][1] += Mi[0];
                dk[i][2] += Mi[1];
                dk[i][4] += Mi[2];
                dk[i][3] = fma(Mi[1], xi, fma(Mi[0], eta, dk[i][3]));
                dk[i][5] = fma(Mi[2], xi, fma(Mi[0], zeta, dk[i][5]));
                dk[i][6] = fma(Mi[2], eta, fma(Mi[1], zeta, dk[i][6]));
                dk[i][7] = fma(Mi[2], xe, fma(Mi[1], xz, fma(Mi[0], ez, dk[i][7])));
            }
        }
    };
    if constexpr (POLY) {
        // J = (c1, c2, c4) of the map, once (what point(0) forms: the mixed coefficients are zero here)
#pragma unroll
        for (int i = 0; i < D; ++i) { J[i][0] = cx[i][1]; J[i][1] = cx[i][2]; J[i][2] = cx[i][4]; }
        const double detJ = det_small<D>(J);
        if (detJ == 0.0) {  // try_inverse fails only for det == 0 exactly (elliptic.rs:401-404)
            if (live) report_singular(a.status, e);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) Ji[i][j] = 0.0;
        } else {
            inv_small(J, detJ, Ji);
        }
        adet = fabs(detJ);
        const double mu = (OP != FH_LAPLACE) ? ep_const(a.qparams)[0] : 0.0, lambda = (OP != FH_LAPLACE) ? ep_const(a.qparams)[1] : 0.0;
        // L restricted: ROWS = which rows (reference directions) of the coefficient matrix are present (bit j: row j = r_j), COLS = which
        // columns of M = P(J^-T R) J^-T are wanted
        auto term = [&](auto rows_tag, auto cols_tag, int c0, int c1, int c2, double (&M)[S][D], double& psi_out) {
            constexpr int ROWS = decltype(rows_tag)::value, COLS = decltype(cols_tag)::value;
            const int cidx[3] = {c0, c1, c2};
            double gu[D][S];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    double t = 0.0;
                    bool first = true;
#pragma unroll
                    for (int m = 0; m < D; ++m)
                        if (ROWS & (1 << m)) {
                            t = first ? Ji[m][i] * cu[k][cidx[m]] : fma(Ji[m][i], cu[k][cidx[m]], t);
                            first = false;
                        }
                    gu[i][k] = t;
                }
            double P[S][D], psi;
            material_point<OP, D, S, WHAT>(gu, mu, lambda, P, psi);
            psi_out = psi;
            if constexpr (WHAT == EP_SCALAR) return;
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int m = 0; m < D; ++m)
                    if (COLS & (1 << m)) {
                        double t = P[i][0] * Ji[m][0];
#pragma unroll
                        for (int k = 1; k < D; ++k) t = fma(P[i][k], Ji[m][k], t);
                        M[i][m] = t;
                    }
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>; using I6 = std::integral_constant<int, 6>;
        using I7 = std::integral_constant<int, 7>;
        // moments of the rule times |det J|: [0] sum w, [1..3] xi^2, eta^2, zeta^2, [4..6] eta^2 zeta^2, xi^2 zeta^2, xi^2 eta^2
        const ep_table mq = ep_const(a.qmom);
        double M1[S][D], Mx[S][D], My[S][D], Mz[S][D], Myz[S][D], Mxz[S][D], Mxy[S][D];
        double p1, px, py, pz, pyz, pxz, pxy;   // the energy density of each part (EP_SCALAR: psi is a quadratic form, the parts do not mix)
        term(I7{}, I7{}, 1, 2, 4, M1, p1);     // constant part: rows c1, c2, c4
        term(I6{}, I6{}, 0, 3, 5, Mx, px);     // coefficient of xi:   d/deta has c3 xi, d/dzeta has c5 xi;   wanted: columns eta, zeta
        term(I5{}, I5{}, 3, 0, 6, My, py);     // coefficient of eta:  d/dxi has c3 eta, d/dzeta has c6 eta;  wanted: columns xi, zeta
        term(I3{}, I3{}, 5, 6, 0, Mz, pz);     // coefficient of zeta: d/dxi has c5 zeta, d/deta has c6 zeta; wanted: columns xi, eta
        term(I1{}, I1{}, 7, 0, 0, Myz, pyz);   // eta zeta: d/dxi has c7;   wanted: column xi
        term(I2{}, I2{}, 0, 7, 0, Mxz, pxz);   // xi zeta:  d/deta has c7;  wanted: column eta
        term(I4{}, I4{}, 0, 0, 7, Mxy, pxy);   // xi eta:   d/dzeta has c7; wanted: column zeta
        const double s0 = mq[0] * adet, sx = mq[1] * adet, sy = mq[2] * adet, sz = mq[3] * adet, syz = mq[4] * adet, sxz = mq[5] * adet,
                     sxy = mq[6] * adet;
        if constexpr (WHAT == EP_SCALAR)
            energy = fma(sxy, pxy, fma(sxz, pxz, fma(syz, pyz, fma(sz, pz, fma(sy, py, fma(sx, px, s0 * p1))))));
#pragma unroll
        for (int i = 0; i < (WHAT == EP_VECTOR ? S : 0); ++i) {
            dk[i][1] = s0 * M1[i][0];
            dk[i][2] = s0 * M1[i][1];
            dk[i][4] = s0 * M1[i][2];
            dk[i][3] = fma(sx, Mx[i][1], sy * My[i][0]);
            dk[i][5] = fma(sx, Mx[i][2], sz * Mz[i][0]);
            dk[i][6] = fma(sy, My[i][2], sz * Mz[i][1]);
            dk[i][7] = fma(sxy, Mxy[i][2], fma(sxz, Mxz[i][1], syz * Myz[i][0]));
        }
    } else if constexpr (AFF) {
        point(0, std::true_type{});
        for (int q = 1; q < a.nq; ++q) point(q, std::false_type{});
    } else if (const_j) {
        point(0, std::true_type{});
        for (int q = 1; q < a.nq; ++q) point(q, std::false_type{});
    } else {
        for (int q = 0; q < a.nq; ++q) point(q, std::true_type{});
    }
    if constexpr (WHAT == EP_VECTOR) {
        // f_n[i] = (1/8) sum_k (sign products of node n over the axes in k) dk[i][k]
#pragma unroll
        for (int i = 0; i < S; ++i) {
            double d[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) d[t] = dk[i][t];
            // synthesis: per axis, (k without the axis: ev, k with it: od) -> node with the bit set ev + od, node with the bit clear ev - od
#pragma unroll
            for (int ax = 1; ax < 8; ax <<= 1)
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (!(b & ax)) {
                        const double ev = d[b], od = d[b | ax];
                        d[b] = ev - od;
                        d[b | ax] = ev + od;
                    }
#pragma unroll
            for (int n = 0; n < N; ++n) f[n][i] = 0.125 * d[hex8_bin(n)];
        }
    } else {
#pragma unroll
        for (int k = 0; k < S; ++k) f[0][k] = 0.0;
    }
}

template <int EK, int OP, int WHAT, int MONO = 0, class UAcc>
__device__ __forceinline__ void element_pass_body(const KArgs& a, const long long e, const bool live, const long long ec,
                                                  const double (&X)[EPDims<EK, OP, WHAT>::N][EPDims<EK, OP, WHAT>::D], const UAcc& U,
                                                  double (&f)[EPDims<EK, OP, WHAT>::NF][EPDims<EK, OP, WHAT>::S], double& energy) {
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    constexpr int D = E::D, N = E::N, S = O::S;
    static_assert(E::NG == N && N <= 8, "element pass: small iso-parametric elements");
    if constexpr (MONO != 0 && EK == FH_HEX8) {   // the monomial form (its own instantiation: both forms in one kernel took 324 registers)
        element_pass_body_hex8<OP, WHAT, MONO == 3 ? 2 : MONO == 2 ? 1 : 0>(a, e, live, ec, X, U, f, energy);
        return;
    }
#pragma unroll
    for (int n = 0; n < (WHAT == EP_VECTOR ? N : 1); ++n)
#pragma unroll
        for (int k = 0; k < S; ++k) f[n][k] = 0.0;
    energy = 0.0;
    const double* par_e = a.rule_map ? a.rparams + (size_t)a.rule_map[ec] * a.nq * 2 : nullptr;
    // The Jacobian of an affine element is the same at every point (simplices always; a Hex8 / Quad4 whose mixed coefficients
    // vanish: every element of a structured, graded or sheared box mesh -- the test of k_classify_affine_hex8):
    // when every element of the wavefront is affine, J, its inverse and the determinant are formed once instead of per point.
    bool const_j = (EK == FH_TET4 || EK == FH_TRI3);
    if constexpr (EK == FH_HEX8 || EK == FH_QUAD4) {
        // reference node signs (hexahedron.rs:49-58, quadrilateral.rs:84-99): the coefficient of xi eta (, xi zeta, eta zeta, xi eta zeta)
        constexpr int SG[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}};
        // against the length of the edges from node 0 (rounding leaves a few ulps of the coordinates in the mixed coefficients of an
        // exact box): 2^-46 like fh_set_affine_tolerance's default -- the Jacobians of such an element differ by less than that
        double len = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) len += fabs(X[1][i] - X[0][i]) + fabs(X[3][i] - X[0][i]) + (D == 3 ? fabs(X[4 % N][i] - X[0][i]) : 0.0);
        const double tol = len * 0x1p-46;
        bool aff = true;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double cxy = 0.0, cxz = 0.0, cyz = 0.0, cxyz = 0.0;
#pragma unroll
            for (int n = 0; n < N; ++n) {
                cxy += (SG[n][0] * SG[n][1]) * X[n][i];
                if (D == 3) {
                    cxz += (SG[n][0] * SG[n][2]) * X[n][i];
                    cyz += (SG[n][1] * SG[n][2]) * X[n][i];
                    cxyz += (SG[n][0] * SG[n][1] * SG[n][2]) * X[n][i];
                }
            }
            aff = aff && fabs(cxy) <= tol && fabs(cxz) <= tol && fabs(cyz) <= tol && fabs(cxyz) <= tol;
        }
        const_j = __all(aff ? 1 : 0) != 0;
    }
    double J[D][D], Ji[D][D], adet = 0.0;
    // one quadrature point; NEED_J (compile time): form J, its inverse and |det J| here, or keep those of the first point
    auto point = [&](int q, auto need_j_tag) {
        constexpr bool need_j = decltype(need_j_tag)::value;
        const ep_table G = ep_const(a.gref) + (size_t)q * N * D;   // uniform over the wavefront: scalar loads
        double R[D][S];
        if constexpr (need_j) {
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) J[i][j] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < S; ++k) R[i][k] = 0.0;
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double g = G[n * D + j];
                if constexpr (need_j) {
#pragma unroll
                    for (int i = 0; i < D; ++i) J[i][j] = fma(X[n][i], g, J[i][j]);   // J = X G^T (hexahedron.rs:101-107)
                }
#pragma unroll
                for (int k = 0; k < S; ++k) R[j][k] = fma(g, U(n, k), R[j][k]);       // sum_n ghat_n u_n^T
            }
        if constexpr (need_j) {
            const double detJ = det_small<D>(J);
            if (detJ == 0.0) {  // try_inverse fails only for det == 0 exactly (elliptic.rs:401-404)
                if (live) report_singular(a.status, e);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) Ji[i][j] = 0.0;
            } else {
                inv_small(J, detJ, Ji);
            }
            adet = fabs(detJ);
        }
        const double s = ep_const(a.qw)[q] * adet;   // w |det J| (elliptic.rs:422)
        double gu[D][S];                          // grad u = J^-T R
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < S; ++k) {
                double t = 0.0;
#pragma unroll
                for (int m = 0; m < D; ++m) t = fma(Ji[m][i], R[m][k], t);
                gu[i][k] = t;
            }
        double mu = 0.0, lambda = 0.0;
        if (OP != FH_LAPLACE) {
            if (par_e) { mu = par_e[2 * q]; lambda = par_e[2 * q + 1]; }
            else { mu = ep_const(a.qparams)[2 * q]; lambda = ep_const(a.qparams)[2 * q + 1]; }
        }
        double P[S][D], psi;
        material_point<OP, D, S, WHAT>(gu, mu, lambda, P, psi);
        if constexpr (WHAT == EP_SCALAR) {
            energy = fma(s, psi, energy);
        } else {
            double M[S][D];   // s P J^-T
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = fma(P[i][k], Ji[m][k], t);
                    M[i][m] = s * t;
                }
#pragma unroll
            for (int n = 0; n < N; ++n)
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    const double g = G[n * D + m];
#pragma unroll
                    for (int i = 0; i < S; ++i) f[n][i] = fma(M[i][m], g, f[n][i]);
                }
        }
    };
    if (const_j) {   // uniform over the wavefront
        point(0, std::true_type{});
        for (int q = 1; q < a.nq; ++q) point(q, std::false_type{});
    } else {
        for (int q = 0; q < a.nq; ++q) point(q, std::true_type{});
    }
}

// WHAT = EP_VECTOR: fe[a][e][c] (a.ke_out, E = a.num_elements);  EP_SCALAR: a.scalar_out[blockIdx.x] = energy of the workgroup's elements
template <int EK, int OP, int WHAT>
__global__ void __launch_bounds__(256) k_element_pass(const KArgs a) {
    constexpr int N = EPDims<EK, OP, WHAT>::N, S = EPDims<EK, OP, WHAT>::S;
    __shared__ double red[4];
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = e < a.num_elements;
    const long long ec = live ? e : a.num_elements - 1;   // every lane computes (uniform control flow), the surplus ones store nothing
    double f[EPDims<EK, OP, WHAT>::NF][S];
    double energy;
    double X[N][EPDims<EK, OP, WHAT>::D], U[N][S];
    element_pass_load<EK, OP, WHAT>(a, ec, X, U);
    element_pass_body<EK, OP, WHAT>(a, e, live, ec, X, EPRegU<N, S>{U}, f, energy);
    if constexpr (WHAT == EP_SCALAR) {
        const double tot = block_sum_256(live ? energy : 0.0, red);
        if (threadIdx.x == 0) a.scalar_out[blockIdx.x] = tot;
    } else {
        if (live) {
#pragma unroll
            for (int n = 0; n < N; ++n) {
                double* dst = a.ke_out + ((size_t)n * (size_t)a.num_elements + (size_t)e) * S;
#pragma unroll
                for (int i = 0; i < S; ++i) dst[i] = f[n][i];
            }
        }
    }
}

// source vector of the small iso-parametric elements (local/source.rs:159-278): f_n = sum_q w |det J| phi_n(xi_q) f(x_q), with
// f = density_q g (GravitySource, fenris-solid/src/gravity_source.rs:57-65) or values sampled by the caller.  fe[a][e][c].
// FACT (GravitySource: f = density_q g with one g for the whole mesh): the element pass leaves the SCALAR  m_n = sum_q w |det J| phi_n rho_q
// per local node (fe[a][e], a third of the scratch traffic of a three-component vector) and the node sum multiplies by g:
// f_n = g sum m_n -- the same numbers as sum_q (w |det J| phi_n)(rho_q g) up to the rounding of the products.
struct SourceG { double v[3]; };   // by value: no device buffer, no copy, no synchronisation per call
// one element's source vector in registers from its vertex coordinates: f[n][k] (FACT: k = 0 only, the scalar m_n)
template <int D, int S, int N, bool FACT>
__device__ __forceinline__ void source_element_body(const KArgs& a, const SourceG& g, const double* values, const long long e, const double (&X)[N][D],
                                                    double (&f)[N][FACT ? 1 : S]) {
    constexpr int SF = FACT ? 1 : S;
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int k = 0; k < SF; ++k) f[n][k] = 0.0;
    double gv[SF];
#pragma unroll
    for (int k = 0; k < SF; ++k) gv[k] = FACT ? 1.0 : (values ? 0.0 : g.v[k]);
    const double* par_e = a.rule_map ? a.rparams + (size_t)a.rule_map[e] * a.nq * 2 : nullptr;
    for (int q = 0; q < a.nq; ++q) {
        const ep_table G = ep_const(a.ggeom) + (size_t)q * N * D;
        double J[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) J[i][j] = 0.0;
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double gg = G[n * D + j];
#pragma unroll
                for (int i = 0; i < D; ++i) J[i][j] = fma(X[n][i], gg, J[i][j]);
            }
        const double wd = ep_const(a.qw)[q] * fabs(det_small<D>(J));
        double fc[SF];
        if (!FACT && values) {
#pragma unroll
            for (int k = 0; k < SF; ++k) fc[k] = values[((size_t)e * a.nq + q) * S + k];
        } else {
            const double rho = par_e ? par_e[2 * q] : (a.qparams ? ep_const(a.qparams)[2 * q] : 0.0);
#pragma unroll
            for (int k = 0; k < SF; ++k) fc[k] = gv[k] * rho;
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const double t = wd * ep_const(a.phiref)[(size_t)q * N + n];
#pragma unroll
            for (int k = 0; k < SF; ++k) f[n][k] = fma(t, fc[k], f[n][k]);
        }
    }
}

template <int D, int S, int N, bool FACT>
__global__ void __launch_bounds__(256) k_source_elements(const KArgs a, const SourceG g, const double* values, double* fe) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.num_elements) return;
    double X[N][D];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int nd = a.conn[(size_t)e * N + n];
#pragma unroll
        for (int i = 0; i < D; ++i) X[n][i] = a.verts[(size_t)nd * D + i];
    }
    constexpr int SF = FACT ? 1 : S;
    double f[N][SF];
    source_element_body<D, S, N, FACT>(a, g, values, e, X, f);
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double* dst = fe + ((size_t)n * (size_t)a.num_elements + (size_t)e) * SF;
#pragma unroll
        for (int k = 0; k < SF; ++k) dst[k] = f[n][k];
    }
}

// second pass: out[s node + c] += sum over the node's (element, local node) entries, ascending entry order (n2e is sorted
// element-major like the reference's sequential loop: bitwise reproducible); entry v = e n + a lives at fe[a][e]
// SO > 0: the entries are scalars (S = 1) and the node's SO components are  g[c] sum  (k_source_elements<FACT>)
template <int S, int SO = 0>
__global__ void __launch_bounds__(256) k_vector_from_elements_soa(int num_nodes, int n, long long E, const unsigned* n2e_off, const unsigned* n2e,
                                                                  const double* fe, double* out, const SourceG g = SourceG{{0.0, 0.0, 0.0}}) {
    static_assert(SO == 0 || S == 1, "scaled node sum: scalar entries");
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= num_nodes) return;
    constexpr int SP = SO > 0 ? SO : S;
    double acc[S], prev[SP];
#pragma unroll
    for (int c = 0; c < S; ++c) acc[c] = 0.0;
#pragma unroll
    for (int c = 0; c < SP; ++c) prev[c] = out[(size_t)node * SP + c];
    const unsigned k0 = n2e_off[node], k1 = n2e_off[node + 1];
    // eight entries at a time: their indices, then all their values, are requested before the first sum (a node of a hexahedral
    // mesh has eight entries: one round trip instead of eight dependent ones); the additions stay in entry order
    for (unsigned kb = k0; kb < k1; kb += 8) {
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = n2e[min(kb + j, k1 - 1)];
        double t[8][S];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned e = v[j] / (unsigned)n, al = v[j] - e * (unsigned)n;
            const double* p = fe + ((size_t)al * (size_t)E + e) * S;
#pragma unroll
            for (int c = 0; c < S; ++c) t[j][c] = p[c];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (kb + j < k1) {
#pragma unroll
                for (int c = 0; c < S; ++c) acc[c] += t[j][c];
            }
    }
    if constexpr (SO > 0) {
#pragma unroll
        for (int c = 0; c < SO; ++c) out[(size_t)node * SO + c] = fma(g.v[c], acc[0], prev[c]);
    } else {
#pragma unroll
        for (int c = 0; c < S; ++c) out[(size_t)node * S + c] = prev[c] + acc[c];
    }
}

// partial sums in index order by one workgroup: thread t takes the partials t, t + 256, ... in order, then the fixed tree
static __global__ void __launch_bounds__(256) k_sum_partials(const double* partial, int n, double* out) {
    __shared__ double red[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    const double tot = block_sum_256(v, red);
    if (threadIdx.x == 0) *out = tot;
}

}  // namespace fenris_hip
