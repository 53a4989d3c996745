// Numeric assembly kernels (matrix / vector / scalar) for gfx950.
//
// One workgroup = one "work unit":
//   element-centric modes (ATOMIC, COLORED, DUMP): `epb` consecutive elements (of a colour list);
//   GATHER (owner-computes): a contiguous range of nodes whose CSR row blocks the workgroup owns.
// Every unit runs the same phases, separated by workgroup barriers:
//   A  stage the unit's unique elements: connectivity, vertex coordinates (and u) -> LDS
//   B  one lane per (element, quadrature point): Jacobian, inverse, |det| w, physical gradients and the
//      operator's per-point coefficients -> LDS   (restates src/assembly/local/elliptic.rs:398-422)
//   C  one lane per (node pair): sum over quadrature points of the operator contraction
//      (operators.rs:146-189, fenris-solid lib.rs:349-392) -> s x s block, then scatter:
//        ATOMIC  : fp64 atomics on the CSR values          COLORED : plain read-modify-write
//        GATHER  : ds_add_f64 into the row accumulators held in LDS, laid out exactly like the CSR rows
//   D  (GATHER) stream the finished rows to HBM once, coalesced.
// The block of a pair (I <= J) is always computed with roles (a = grad phi_I, b = grad phi_J) and the
// (J, I) block is its transpose -- the same values clone_upper_to_lower produces (src/util.rs:38-51).
#pragma once
#include <type_traits>
#include "device_common.hpp"
#include "small_ops.hpp"

namespace fenris_hip {

// ------------------------------------------------------------------------------------------ LDS carve
struct Layout {
    int o_gref, o_ggeom, o_qw, o_qpar, o_X, o_U, o_QP, o_ACC, n_doubles;
    int o_uniq, o_cn, o_ent, o_ncols, o_noff, o_nc, o_ncr, n_ints;
    int qpd;   // doubles per (element, quadrature point), padded to an odd count (LDS bank spread)
    int fast;  // 1: gradients stored pre-scaled by sqrt(w |det J|), no per-point coefficients
    int o_pos; // gather: per (entry, local node) column slot, bytes
    int nqs;   // quadrature points staged per element at a time (== nq unless chunked)
    int qss;   // doubles per staged element (nqs * qpd, plus padding in the planar form)
    int planar;  // 1: gradients of a point stored component-major ([c][node], 16-byte aligned rows) for ds_read_b128
    // strides in doubles, rounded up to odd counts: phase B runs one lane per (element, point), so lanes differ
    // in the element (X, U) or in the point (gradient tables); an even stride such as 24 maps eight of them onto
    // two LDS bank pairs (4-way conflicts on three quarters of phase B's LDS traffic)
    int xs, us, gs, ggs;  // per element: vertices, u;  per point: reference gradients, geometry gradients
    __host__ __device__ size_t bytes() const { return sizeof(double) * (size_t)n_doubles + sizeof(int) * (size_t)n_ints; }
};

enum { WHAT_MATRIX = 0, WHAT_VECTOR = 1, WHAT_SCALAR = 2 };

template <int EK, int OP, int WHAT>
__host__ __device__ inline Layout make_layout(int nq, int ub, int acc_max, int nb_max, bool gather, int mb = 0, int fast = 0,
                                               int nq_stage = 0, int nc_row = 0, int planar = 0) {
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    Layout L;
    int o = 0;
    L.fast = (WHAT == WHAT_MATRIX && (OP == FH_LAPLACE || OP == FH_LINEAR_ELASTIC)) ? fast : 0;
    L.planar = (planar && L.fast) ? 1 : 0;
    L.xs = (E::NG * E::D) | 1;
    L.us = (E::N * O::S) | 1;
    L.gs = (E::N * E::D) | 1;
    L.ggs = (E::NG * E::D) | 1;
    if (L.planar) {
        // phase B reads whole rows as ds_read_b128 pairs: even strides (16-byte aligned rows); 2 (n + 1) doubles per
        // row puts the rows of eight points, and of nearby slots, on different bank quads
        L.xs = E::NG * E::D + 2;
        L.gs = E::N * E::D + 2;
        L.ggs = E::NG * E::D + 2;
    }
    L.o_gref = o;  o += nq * L.gs;
    if (E::NG == E::N) L.o_ggeom = L.o_gref;  // iso-parametric: one table
    else { L.o_ggeom = o; o += nq * L.ggs; }
    L.o_qw = o;    o += nq;
    L.o_qpar = o;  o += 2 * nq;
    if (L.planar) o += o & 1;
    L.o_X = o;     o += ub * L.xs;
    L.o_U = o;     o += (O::NEEDS_U || WHAT != WHAT_MATRIX) ? ub * L.us : 0;
    if (WHAT == WHAT_MATRIX) L.qpd = O::NVEC * E::N * E::D + (L.fast ? 0 : O::NCOEF);
    else if (WHAT == WHAT_VECTOR) L.qpd = (fast ? 0 : E::N * E::D) + O::S * E::D;  // fast: the VCOMPACT record of prologue()
    else L.qpd = 1;
    if (WHAT != WHAT_SCALAR && (L.qpd & 1) == 0) L.qpd += 1;
    L.nqs = nq_stage > 0 ? nq_stage : nq;  // quadrature points staged at a time
    L.qss = L.nqs * L.qpd;
    if (L.planar && planar == 2) {
        // node-major rows padded to 4 doubles per node ([x y | z 0]: two ds_read_b128 per vector) for the row-owner
        // kernel (rows_kernel.hpp); 4 N + 2 doubles per row: the 8 points of a slot land on disjoint bank quads for
        // phase B's ds_write_b128
        L.qpd = 4 * E::N + 2;
        L.qss = L.nqs * L.qpd;
        o += o & 1;
    } else if (L.planar) {
        // rows of 2 (N D / 2 + 1) doubles: even (16-byte alignment of every row), and 8 consecutive points of a slot
        // land on disjoint bank quads for phase B's ds_write_b128; slot stride = 8 (mod 16) doubles: the 64-byte
        // footprints that phase C's 16-lane ds_read_b128 groups fetch from four different slots cover all 64 banks
        L.qpd = E::N * E::D + 2;
        L.qss = L.nqs * L.qpd;
        L.qss += (8 - L.qss % 16 + 16) % 16;
        o += o & 1;
    }
    L.o_QP = o;    o += ub * L.qss;
    L.o_ACC = o;   o += gather ? acc_max : 0;
    L.n_doubles = o;
    int i = 0;
    L.o_uniq = i;    i += gather ? mb : ub;   // gather: all unique elements of an entry batch
    L.o_cn = i;      i += ub * E::N;
    L.o_ent = i;     i += gather ? mb : 0;
    L.o_ncols = i;   i += gather ? acc_max / (O::S * O::S) : 0;
    L.o_noff = i;    i += gather ? nb_max + 1 : 0;
    L.o_pos = i;     i += gather ? (mb * E::N + 3) / 4 : 0;
    // element-centric kernels: neighbour lists of the staged elements' nodes (nc_row = longest list) so that the
    // column search of the scatter runs in LDS instead of dependent global loads
    L.o_ncr = i;     i += (!gather && nc_row > 0) ? 2 * ub * E::N : 0;
    L.o_nc = i;      i += (!gather && nc_row > 0) ? ub * E::N * nc_row : 0;
    L.n_ints = i + 4;
    // pipelined gather (the only kernel that stages quadrature points in chunks) lays out its integers itself:
    // 16 header words + two parity copies of (slot list | entries | column slots | row offsets)
    if (gather && nq_stage > 0) L.n_ints = 2 * (8 + ub / 4 + mb + (mb * E::N + 3) / 4 + nb_max + 1) + 8;
    return L;
}



// compile-time loop 0..N-1: f(std::integral_constant<int, k>) -- the explicit LDS waits need immediate operands
template <int N, int D, int K = 0, typename F>
__device__ __forceinline__ void pipeline_consume(F&& f) {
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        pipeline_consume<N, D, K + 1>(f);
    }
}

// ------------------------------------------------------------------------------------------ phase B
// One lane per (staged element u, quadrature point q).  Writes the LDS record qp[] described by OpT.
// VCOMPACT (WHAT_VECTOR only): the point record is the 3 x 3 matrix  M = s P J^-T  instead of the N physical gradients and
// s P -- the element vector is  f_n = sum_q M_q grad_ref_n(xi_q)  with the reference gradients from the (constant) table;
// grad u comes from  J^-T (sum_n grad_ref_n u_n^T), so the physical gradients are never formed
// XIND (planar Tet4, rows_kernel.hpp): the vertices of the element are not a row of X but four entries of a table of the position's
//   UNIQUE vertices (32 bytes each: [x y | z -]) at L.o_X, selected by the four bytes of `xind`
template <int EK, int OP, int WHAT, bool PLANAR = false, bool NODEMAJOR = false, bool VCOMPACT = false, bool XIND = false>
__device__ __forceinline__ void prologue(const KArgs& a, const Layout& L, double* lds, const int* lds_i, int u, int q,
                                         const int* elem_id, int qslot = -1, double sqw = 0.0, unsigned xind = 0u) {
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    constexpr int D = E::D, N = E::N, NG = E::NG, S = O::S;
    constexpr bool IS_MASS = (OP == FH_MASS_SCALAR || OP == FH_MASS_VECTOR);
    const double* X = lds + L.o_X + u * L.xs;
    const double* gg = lds + L.o_ggeom + q * L.ggs;
    const double* gr = lds + L.o_gref + q * L.gs;
    double* qp = lds + L.o_QP + (size_t)u * L.qss + (size_t)(qslot < 0 ? q : qslot) * L.qpd;
    (void)lds_i;

    // J = X G^T  (hexahedron.rs:101-107): J[i][j] = sum_g x_g[i] dN_g/dxi_j
    double J[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) J[i][j] = 0.0;
    constexpr int UNR = (N <= 8) ? N : 3;
    constexpr bool EXPLICIT_LDS = (N <= 8);  // hand-issued ds_read_b64 (see lds_read_f64): small elements only
    // iso-parametric small elements: the rows of the gradient table fetched for J stay in registers for the
    // physical gradients below (a quarter of phase B's LDS traffic)
    constexpr bool KEEP_G = EXPLICIT_LDS && NG == N && !IS_MASS;
    double gkeep[KEEP_G ? N : 1][D];
    if constexpr (PLANAR) {
        // rows of X and of the gradient table as ds_read_b128 pairs (flat index 3 g + c), two nodes (3 + 3 fetches)
        // per step, two steps in flight: 24 wide fetches instead of 48 narrow ones
        static_assert(!PLANAR || (NG == N && N % 2 == 0 && D == 3), "planar phase B: iso-parametric, even node count, 3D");
        const unsigned xa = (unsigned)(unsigned long long)X, ga = (unsigned)(unsigned long long)gg;
        const unsigned xtab = (unsigned)(unsigned long long)(lds + L.o_X);
        constexpr int NS = N / 2;  // steps
        f64x2 xf[NS][3], gf[NS][3];
        double xz[XIND ? NS : 1][2];
        auto fetch2 = [&](auto sk) {
            constexpr int st = decltype(sk)::value;
            if constexpr (XIND) {
                const unsigned a0 = xtab + 32u * ((xind >> (16 * st)) & 255u), a1 = xtab + 32u * ((xind >> (16 * st + 8)) & 255u);
                xf[st][0] = lds_read_f64x2<0>(a0);
                xz[st][0] = lds_read_f64_at<16>(a0);
                xf[st][1] = lds_read_f64x2<0>(a1);
                xz[st][1] = lds_read_f64_at<16>(a1);
            } else {
                xf[st][0] = lds_read_f64x2<(6 * st) * 8>(xa);
                xf[st][1] = lds_read_f64x2<(6 * st + 2) * 8>(xa);
                xf[st][2] = lds_read_f64x2<(6 * st + 4) * 8>(xa);
            }
            gf[st][0] = lds_read_f64x2<(6 * st) * 8>(ga);
            gf[st][1] = lds_read_f64x2<(6 * st + 2) * 8>(ga);
            gf[st][2] = lds_read_f64x2<(6 * st + 4) * 8>(ga);
        };
        constexpr int PER = XIND ? 7 : 6;   // LDS fetches of a step
        fetch2(std::integral_constant<int, 0>{});
        fetch2(std::integral_constant<int, 1>{});
        pipeline_consume<NS, D>([&](auto sk) {
            constexpr int st = decltype(sk)::value;
            lds_wait<(st + 1 < NS) ? PER : 0>();
            if constexpr (st + 2 < NS) fetch2(std::integral_constant<int, st + 2>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (XIND) asm volatile("" : "+v"(xz[XIND ? st : 0][0]), "+v"(xz[XIND ? st : 0][1]));
            const double xe[2][3] = {{xf[st][0].x, xf[st][0].y, XIND ? xz[XIND ? st : 0][0] : xf[st][1].x},
                                     {XIND ? xf[st][1].x : xf[st][1].y, XIND ? xf[st][1].y : xf[st][2].x, XIND ? xz[XIND ? st : 0][1] : xf[st][2].y}};
            const double ge[2][3] = {{gf[st][0].x, gf[st][0].y, gf[st][1].x}, {gf[st][1].y, gf[st][2].x, gf[st][2].y}};
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    gkeep[KEEP_G ? 2 * st + h : 0][i] = ge[h][i];
#pragma unroll
                    for (int j = 0; j < D; ++j) J[i][j] = fma(xe[h][i], ge[h][j], J[i][j]);
                }
        });
    } else if (EXPLICIT_LDS && KEEP_G) {
        // software-pipelined two nodes ahead: 4 D fetches (<= 12 of the 15 LDS operations the lgkm counter can track)
        // are in flight while node g is accumulated; LDS returns in order
        double xall[NG][D];
        constexpr int AHEAD = (NG > 2) ? 2 : 1;
#pragma unroll
        for (int g = 0; g < AHEAD; ++g) {
            lds_read_vec<D>(X + g * D, xall[g]);
            lds_read_vec<D>(gg + g * D, gkeep[KEEP_G ? g : 0]);
        }
        pipeline_consume<NG, D>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + AHEAD < NG) {
                lds_read_vec<D>(X + (g + AHEAD) * D, xall[g + AHEAD]);
                lds_read_vec<D>(gg + (g + AHEAD) * D, gkeep[KEEP_G ? g + AHEAD : 0]);
            }
            constexpr int pending = ((NG - 1 - g) < AHEAD ? (NG - 1 - g) : AHEAD) * 2 * D;
            lds_wait<pending>();
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) J[i][j] = fma(xall[g][i], gkeep[KEEP_G ? g : 0][j], J[i][j]);
        });
    } else if (EXPLICIT_LDS) {
        // software-pipelined: the fetches of node g+1 are in flight while node g is accumulated
        double xb[2][D], gb[2][D];
        lds_read_vec<D>(X, xb[0]);
        lds_read_vec<D>(gg, gb[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
                lds_read_vec<D>(X + (g + 1) * D, xb[(g + 1) & 1]);
                lds_read_vec<D>(gg + (g + 1) * D, gb[(g + 1) & 1]);
                lds_wait<2 * D>();
            } else {
                lds_wait<0>();
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) J[i][j] = fma(xb[g & 1][i], gb[g & 1][j], J[i][j]);
        }
    } else {
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) J[i][j] = fma(X[g * D + i], gg[g * D + j], J[i][j]);
    }
    const double detJ = det_small<D>(J);
    double Ji[D][D];
    if (detJ == 0.0) {  // try_inverse fails only for det == 0 exactly (elliptic.rs:401-404)
        // the mass assembler only uses |det J| (mass.rs:245): a degenerate element contributes zero, no error
        if (!IS_MASS) report_singular(a.status, (long long)*elem_id);  // dereferenced only on this (rare) path
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) Ji[i][j] = 0.0;
    } else if constexpr (PLANAR) {
        // sqrt(w |det J|) J^-1 = sign(det J) sqrt(w) / sqrt(|det J|) adj(J): one reciprocal square root instead of a
        // division, a square root and the scaling of the gradients (sqw = sqrt(w_q), constant per lane)
        adj_scaled(J, copysign(sqw, detJ) * rsqrt_newton(fabs(detJ)), Ji);
    } else {
        inv_small(J, detJ, Ji);
    }
    const double s = lds[L.o_qw + q] * fabs(detJ);  // scale = w |det J| (elliptic.rs:422)

    // physical gradients g_n = J^{-T} grad_ref_n ; optionally grad u = sum_n g_n u_n^T  (d x s)
    constexpr bool want_u = O::NEEDS_U || WHAT != WHAT_MATRIX;
    double gu[D][S];
    if (want_u) {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < S; ++k) gu[i][k] = 0.0;
    }
    const double* Ue = lds + L.o_U + u * L.us;
    double* gout = qp;  // first node-vector block: physical gradients
    // FAST path (Laplace / uniform linear elasticity, non-negative weights): store sqrt(s) g_n so that
    // sum_q s g_I g_J^T = sum_q h_I h_J^T needs no per-point coefficient in phase C
    const double fast_scale = (WHAT == WHAT_MATRIX && L.fast && !PLANAR) ? sqrt(s) : 1.0;
    if (IS_MASS) {
        // mass.rs:243-270: only |det J|, the density and the basis values enter; phi_n goes to the first component
        if (WHAT == WHAT_MATRIX) {
            for (int n = 0; n < N; ++n) {
                gout[n * D] = a.phiref[q * N + n];
#pragma unroll
                for (int i = 1; i < D; ++i) gout[n * D + i] = 0.0;
            }
            qp[O::NVEC * N * D] = s * (a.rule_map ? a.rparams[((size_t)a.rule_map[*elem_id] * a.nq + q) * 2] : lds[L.o_qpar + 2 * q]);
        }
        return;
    }
    double rb[2][D];
    double Rm[VCOMPACT ? D : 1][VCOMPACT ? S : 1];  // VCOMPACT: sum_n grad_ref_n u_n^T
    if (VCOMPACT) {
#pragma unroll
        for (int m = 0; m < D; ++m)
#pragma unroll
            for (int k = 0; k < S; ++k) Rm[m % (VCOMPACT ? D : 1)][k % (VCOMPACT ? S : 1)] = 0.0;
    }
    double gpair[D];  // PLANAR: the even node of a pair waits here for its odd neighbour
    if (EXPLICIT_LDS && !KEEP_G) lds_read_vec<D>(gr, rb[0]);
#pragma unroll UNR
    for (int n = 0; n < N; ++n) {
        double g[D], rv[D];
        if (KEEP_G) {
#pragma unroll
            for (int k = 0; k < D; ++k) rv[k] = gkeep[KEEP_G ? n : 0][k];
        } else if (EXPLICIT_LDS) {
            if (n + 1 < N) {
                lds_read_vec<D>(gr + (n + 1) * D, rb[(n + 1) & 1]);
                lds_wait<D>();
            } else {
                lds_wait<0>();
            }
#pragma unroll
            for (int k = 0; k < D; ++k) rv[k] = rb[n & 1][k];
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) rv[k] = gr[n * D + k];
        }
        if constexpr (VCOMPACT) {
#pragma unroll
            for (int m = 0; m < D; ++m)
#pragma unroll
                for (int k = 0; k < S; ++k) Rm[m][k] = fma(rv[m], Ue[n * S + k], Rm[m][k]);
            continue;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t = fma(Ji[k][i], rv[k], t);
            g[i] = t;
        }
        if (PLANAR && NODEMAJOR) {
            // node-major rows padded to four doubles: [x y | z 0]
            f64x2 lo, hi;
            lo.x = g[0]; lo.y = g[1 % D];
            hi.x = g[2 % D]; hi.y = 0.0;
            *reinterpret_cast<f64x2*>(gout + 4 * n) = lo;
            *reinterpret_cast<f64x2*>(gout + 4 * n + 2) = hi;
        } else if (PLANAR) {
            // component-major rows [c][node]: the values of nodes (n - 1, n) leave as one ds_write_b128 per component
            if (n & 1) {
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    f64x2 pr;
                    pr.x = gpair[i];
                    pr.y = g[i];
                    *reinterpret_cast<f64x2*>(gout + i * N + (n - 1)) = pr;
                }
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i) gpair[i] = g[i];
            }
        } else if (WHAT != WHAT_SCALAR) {
#pragma unroll
            for (int i = 0; i < D; ++i) gout[n * D + i] = fast_scale * g[i];
        }
        if (want_u) {
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < S; ++k) gu[i][k] = fma(g[i], Ue[n * S + k], gu[i][k]);
        }
    }
    if constexpr (VCOMPACT) {  // grad u = J^-T R
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < S; ++k) {
                double t = 0.0;
#pragma unroll
                for (int m = 0; m < D; ++m) t = fma(Ji[m][i], Rm[m % (VCOMPACT ? D : 1)][k % (VCOMPACT ? S : 1)], t);
                gu[i][k] = t;
            }
    }
    double mu = 0.0, lambda = 0.0;
    if (OP != FH_LAPLACE) {
        if (a.rule_map) {  // per-element data (compact table): the slow path, never with L.fast
            const double* par = a.rparams + ((size_t)a.rule_map[*elem_id] * a.nq + q) * 2;
            mu = par[0];
            lambda = par[1];
            // landed inside this branch: left to the merge point, the wait for this fetch is placed on the common path as
            // vmcnt(0), where it also waits for the register prefetch of the streaming kernels' next batch
            asm volatile("" : "+v"(mu), "+v"(lambda));
        } else {
            // through an LDS pointer proper: with a generic one the two branches become one flat load of a selected address,
            // which counts as a vector-memory operation as well (same vmcnt(0))
            typedef const __attribute__((address_space(3))) double* lds_cptr;
            const lds_cptr qpar = (lds_cptr)(lds + L.o_qpar);
            mu = qpar[2 * q];
            lambda = qpar[2 * q + 1];
        }
    }

    // deformation gradient F = I + (grad u)^T  (fenris-solid/src/lib.rs:20-29)
    double F[D][D];
    if (S == D && want_u) {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) F[i][j] = (i == j ? 1.0 : 0.0) + gu[j][i % S];
    }

    if (WHAT == WHAT_MATRIX) {
        double* coef = qp + O::NVEC * N * D;
        if (L.fast) {
            // nothing: scale folded into the gradients
        } else if (OP == FH_LAPLACE || OP == FH_TENSOR) {
            coef[0] = s;
        } else if (OP == FH_LINEAR_ELASTIC) {
            coef[0] = s * mu;
            coef[1] = s * lambda;
        } else if (OP == FH_NEO_HOOKEAN) {
            // materials.rs:287-315: J <= 0 => all-NaN block
            const double Jd = det_small<D>(F);
            double Fi[D][D];
            double c_l, c_a, c_m;
            if (Jd <= 0.0) {
                c_l = c_a = c_m = __builtin_nan("");
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) Fi[i][j] = 0.0;
            } else {
                inv_small(F, Jd, Fi);
                const double alpha = -mu + lambda * log(Jd);
                c_l = s * lambda; c_a = s * alpha; c_m = s * mu;
            }
            coef[0] = c_l; coef[1] = c_a; coef[2] = c_m;
            double* at = qp + N * D;  // F^{-T} g_n
            for (int n = 0; n < N; ++n)
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = fma(Fi[k][i], gout[n * D + k], t);
                    at[n * D + i] = t;
                }
        } else {  // StVK, materials.rs:417-438
            double Eg[D][D];  // Green strain (F^T F - I)/2
            double trE = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = fma(F[k][i], F[k][j], t);
                    Eg[i][j] = (t - (i == j ? 1.0 : 0.0)) * 0.5;
                }
#pragma unroll
            for (int i = 0; i < D; ++i) trE += Eg[i][i];
            coef[0] = s * 2.0 * mu;
            coef[1] = s * lambda * trE;
            coef[2] = s * mu;
            coef[3] = s * lambda;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = fma(F[i][k], F[j][k], t);
                    coef[4 + i * D + j] = t;  // F F^T
                }
            double* fg = qp + N * D;       // F g_n
            double* eg = qp + 2 * N * D;   // E g_n
            for (int n = 0; n < N; ++n)
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double t = 0.0, t2 = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        t = fma(F[i][k], gout[n * D + k], t);
                        t2 = fma(Eg[i][k], gout[n * D + k], t2);
                    }
                    fg[n * D + i] = t;
                    eg[n * D + i] = t2;
                }
        }
    } else {
        // stress P (s x d), scaled by s, for the residual; energy density for the scalar path
        double P[S][D];
        double psi = 0.0;
        if (OP == FH_LAPLACE) {
            // g^T = (grad u)^T (laplace.rs:52-56); psi = 1/2 |grad u|^2 (laplace.rs:35-37)
#pragma unroll
            for (int k = 0; k < D; ++k) { P[0][k] = gu[k][0]; psi = fma(gu[k][0], gu[k][0], psi); }
            psi *= 0.5;
        } else if (OP == FH_LINEAR_ELASTIC) {
            // eps = sym(F) - I formed from F like the reference does (materials.rs:71-79)
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
                for (int j = 0; j < D; ++j) P[i % S][j] = eps[i][j] * 2.0 * mu + (i == j ? lambda * tr : 0.0);
            psi = mu * ee + 0.5 * lambda * (tr * tr);
        } else if (OP == FH_NEO_HOOKEAN) {
            const double Jd = det_small<D>(F);
            if (Jd <= 0.0) {  // materials.rs:271-274
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) P[i % S][j] = __builtin_nan("");
            } else {
                double Fi[D][D];
                inv_small(F, Jd, Fi);
                const double c = -mu + lambda * log(Jd);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) P[i % S][j] = Fi[j][i] * c + F[i][j] * mu;
            }
            if (WHAT == WHAT_SCALAR) {
                // materials.rs:249-262 with log_det_F of du_dX = (grad u)^T (logdet.rs:17-86)
                double U[D][D];
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) U[i][j] = gu[j][i % S];
                double gamma;
                if (D == 2) {
                    gamma = U[0][0] * U[1][1] + U[0][0] + U[1][1] - U[0][1] * U[1][0];
                } else {
                    const double u11 = U[0][0], u22 = U[1][1], u33 = U[2 % D][2 % D];
                    const double aa = 1.0 + u11, e2 = 1.0 + u22, i2 = 1.0 + u33;
                    const double b = U[0][1], c = U[0][2 % D], d2 = U[1][0], f = U[1][2 % D], g = U[2 % D][0], h = U[2 % D][1];
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
                    P[i % S][j] = t * 2.0 * mu + F[i][j] * lambda * trE;
                }
            psi = mu * ee + 0.5 * lambda * (trE * trE);
        }
        if (WHAT == WHAT_VECTOR && VCOMPACT) {
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) t = fma(P[i][k], Ji[m][k], t);
                    qp[i * D + m] = s * t;
                }
        } else if (WHAT == WHAT_VECTOR) {
            double* sp = qp + N * D;
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) sp[i * D + k] = s * P[i][k];
        } else {
            qp[0] = s * psi;
        }
    }
}


// ------------------------------------------------------------------------------------------ phase C
// s x s block K_e[(I,.),(J,.)] for I <= J: sum over quadrature points of scale * C(grad u; g_I, g_J).
template <int EK, int OP>
__device__ __forceinline__ void pair_block(const KArgs& ka, const Layout& L, const double* lds, int nq, int u, int I, int J,
                                           double (&blk)[OpT<OP, ElemT<EK>::D>::S][OpT<OP, ElemT<EK>::D>::S]) {
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    constexpr int D = E::D, N = E::N, S = O::S;
    const double* qp = lds + L.o_QP + (size_t)u * nq * L.qpd;
    if ((OP == FH_LAPLACE || OP == FH_LINEAR_ELASTIC) && L.fast) {
        // G = sum_q h_I h_J^T with h = sqrt(w |det J|) grad phi
        double G[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) G[i][j] = 0.0;
        const double* a = qp + I * D;
        const double* b = qp + J * D;
        for (int q = 0; q < nq; ++q, a += L.qpd, b += L.qpd) {
            double av[D], bv[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { av[i] = a[i]; bv[i] = b[i]; }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) G[i][j] = fma(av[i], bv[j], G[i][j]);
        }
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) tr += G[i][i];
        if (OP == FH_LAPLACE) {
            blk[0][0] = tr;
        } else {
            // mu [(a.b) I + b a^T] + lambda a b^T  summed:  mu (tr G I + G^T) + lambda G
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j)
                    blk[i % S][j % S] = ka.mu * ((i == j ? tr : 0.0) + G[j][i]) + ka.lambda * G[i][j];
        }
    } else if (OP == FH_MASS_SCALAR || OP == FH_MASS_VECTOR) {
        double m = 0.0;  // sum_q (s rho) phi_I phi_J ; block = m I_s (mass.rs:262-270)
        for (int q = 0; q < nq; ++q, qp += L.qpd) m = fma(qp[N * D] * qp[I * D], qp[J * D], m);
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int j = 0; j < S; ++j) blk[i][j] = (i == j) ? m : 0.0;
    } else if (OP == FH_TENSOR) {
        // C(a, b)[i][k] = sum_jl a[j] A[i][j][k][l] b[l] with the point's tensor (fenris_hip.h, FH_TENSOR; what `contract` of operators.rs:146-161
        // returns for a gradient-independent operator), scaled by w |det J| (elliptic.rs:422): the block of the ORDERED pair (I, J)
        double B[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) B[i][k] = 0.0;
        for (int q = 0; q < nq; ++q, qp += L.qpd) {
            const double* a = qp + I * D;
            const double* b = qp + J * D;
            const double* A = ka.tensor + (size_t)q * (D * D * D * D);
            const double sc = qp[N * D];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double t = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j)
#pragma unroll
                        for (int l = 0; l < D; ++l) t = fma(a[j] * A[((i * D + j) * D + k) * D + l], b[l], t);
                    B[i][k] = fma(sc, t, B[i][k]);
                }
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) blk[i % S][k % S] = B[i][k];
    } else if (OP == FH_LAPLACE) {
        double k = 0.0;
        for (int q = 0; q < nq; ++q, qp += L.qpd) {
            const double* a = qp + I * D;
            const double* b = qp + J * D;
            double dt = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) dt = fma(a[i], b[i], dt);
            k = fma(qp[N * D], dt, k);
        }
        blk[0][0] = k;
    } else if (OP == FH_LINEAR_ELASTIC) {
        // C = mu [(a.b) I + b a^T] + lambda a b^T (materials.rs:108-118), per-point coefficients:
        // M1 = sum (s mu a) b^T, M2 = sum (s lambda a) b^T  =>  block = tr(M1) I + M1^T + M2
        double M1[D][D], M2[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) { M1[i][j] = 0.0; M2[i][j] = 0.0; }
        for (int q = 0; q < nq; ++q, qp += L.qpd) {
            const double* a = qp + I * D;
            const double* b = qp + J * D;
            const double cm = qp[N * D], cl = qp[N * D + 1];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double am = cm * a[i], al = cl * a[i];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    M1[i][j] = fma(am, b[j], M1[i][j]);
                    M2[i][j] = fma(al, b[j], M2[i][j]);
                }
            }
        }
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) tr += M1[i][i];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) blk[i % S][j % S] = (i == j ? tr : 0.0) + M1[j][i] + M2[i][j];
    } else if (OP == FH_NEO_HOOKEAN) {
        // lambda (F^-T a)(F^-T b)^T - alpha (F^-T b)(F^-T a)^T + mu (a.b) I   (materials.rs:302-313)
        double B[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) B[i][j] = 0.0;
        for (int q = 0; q < nq; ++q, qp += L.qpd) {
            const double* a = qp + I * D;
            const double* b = qp + J * D;
            const double* ta = qp + N * D + I * D;
            const double* tb = qp + N * D + J * D;
            const double* c = qp + 2 * N * D;
            double dt = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) dt = fma(a[i], b[i], dt);
            const double diag = c[2] * dt;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double tl = c[0] * ta[i], tal = c[1] * tb[i];
#pragma unroll
                for (int j = 0; j < D; ++j) B[i][j] += tl * tb[j] - tal * ta[j] + (i == j ? diag : 0.0 * diag);
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) blk[i % S][j % S] = B[i][j];
    } else {  // StVK (materials.rs:417-438)
        double B[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) B[i][j] = 0.0;
        for (int q = 0; q < nq; ++q, qp += L.qpd) {
            const double* a = qp + I * D;
            const double* b = qp + J * D;
            const double* fa = qp + N * D + I * D;
            const double* fb = qp + N * D + J * D;
            const double* eb = qp + 2 * N * D + J * D;
            const double* c = qp + 3 * N * D;
            double ab = 0.0, aeb = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) { ab = fma(a[i], b[i], ab); aeb = fma(a[i], eb[i], aeb); }
            const double diag = c[0] * aeb + c[1] * ab;
            const double mab = c[2] * ab;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j)
                    B[i][j] += (i == j ? diag : 0.0) + c[2] * fb[i] * fa[j] + c[3] * fa[i] * fb[j] + mab * c[4 + i * D + j];
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) blk[i % S][j % S] = B[i][j];
    }
    if (I == J && !(OP == FH_TENSOR && (ka.nonsym & 1))) {  // scalar-level mirror inside the diagonal block (util.rs:46-50; not for Symmetry::NonSymmetric)
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int j = 0; j < S; ++j)
                if (i > j) blk[i][j] = blk[j][i];
    }
}

// position of column node j inside the (ascending) neighbour list of a row node
__device__ __forceinline__ int find_col(const unsigned* cols, int cnt, unsigned j) {
    int lo = 0, hi = cnt;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cols[mid] < j) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int find_col_lds(const int* cols, int cnt, int j) {
    int lo = 0, hi = cnt;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cols[mid] < j) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int MODE> __device__ __forceinline__ void add_value(double* p, double v) {
    if (MODE == MODE_ATOMIC) atomic_add_f64(p, v);
    else *p += v;  // colours are disjoint: plain read-modify-write (paradis lib.rs:258-278)
}

// stage tables shared by all phases
template <int EK>
__device__ __forceinline__ void stage_tables(const KArgs& a, const Layout& L, double* lds) {
    using E = ElemT<EK>;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < a.nq * E::N * E::D; i += nt) lds[L.o_gref + (i / (E::N * E::D)) * L.gs + i % (E::N * E::D)] = a.gref[i];
    if (E::NG != E::N)
        for (int i = tid; i < a.nq * E::NG * E::D; i += nt)
            lds[L.o_ggeom + (i / (E::NG * E::D)) * L.ggs + i % (E::NG * E::D)] = a.ggeom[i];
    for (int i = tid; i < a.nq; i += nt) lds[L.o_qw + i] = a.qw[i];
    for (int i = tid; i < 2 * a.nq; i += nt) lds[L.o_qpar + i] = a.qparams ? a.qparams[i] : 0.0;
}

// stage connectivity, vertices and u of the U unique elements listed in lds_i[o_uniq..]
template <int EK, int S>
__device__ __forceinline__ void stage_elements(const KArgs& a, const Layout& L, double* lds, int* lds_i, int U, bool need_u,
                                               const int* uniq) {
    using E = ElemT<EK>;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < U * E::N; i += nt) {
        const int u = i / E::N, n = i % E::N;
        lds_i[L.o_cn + i] = a.conn[(size_t)uniq[u] * E::N + n];
    }
    __syncthreads();
    for (int i = tid; i < U * E::NG * E::D; i += nt) {
        const int u = i / (E::NG * E::D), r = i % (E::NG * E::D), g = r / E::D, c = r % E::D;
        lds[L.o_X + u * L.xs + r] = a.verts[(size_t)lds_i[L.o_cn + u * E::N + g] * E::D + c];
    }
    if (need_u) {
        for (int i = tid; i < U * E::N * S; i += nt) {
            const int u = i / (E::N * S), r = i % (E::N * S), n = r / S, c = r % S;
            lds[L.o_U + u * L.us + r] = a.u ? a.u[(size_t)lds_i[L.o_cn + u * E::N + n] * S + c] : 0.0;
        }
    }
    __syncthreads();
}

// triangular pair index p -> (I <= J), p = J (J+1)/2 + I
__device__ __forceinline__ void unpack_pair(int p, int& I, int& J) {
    int j = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
    while ((j + 1) * (j + 2) / 2 <= p) ++j;
    while (j * (j + 1) / 2 > p) --j;
    J = j;
    I = p - j * (j + 1) / 2;
}

// element-centric store of a NON-SYMMETRIC pair of blocks (FH_TENSOR): p1 goes to (I, J) as it is, q1 TRANSPOSED to (J, I) -- the two stores of
// k_assemble_matrix below with separate sources
template <int EK, int OP, int MODE>
__device__ __forceinline__ void tensor_store(const KArgs& a, const Layout& L, const int* lds_i, bool nc_lds, int u, int I, int J, long long w0,
                                             const double (&p1)[OpT<OP, ElemT<EK>::D>::S][OpT<OP, ElemT<EK>::D>::S],
                                             const double (&q1)[OpT<OP, ElemT<EK>::D>::S][OpT<OP, ElemT<EK>::D>::S]) {
    using E = ElemT<EK>;
    constexpr int N = E::N, S = OpT<OP, E::D>::S;
    if (MODE == MODE_DUMP) {
        double* ke = a.ke_out + (size_t)(a.ke_by_elem ? (long long)lds_i[L.o_uniq + u] : (w0 - a.work_begin + u)) * (S * N) * (S * N);
        for (int j = 0; j < S; ++j)
            for (int i = 0; i < S; ++i) ke[(size_t)(S * J + j) * (S * N) + S * I + i] = p1[i][j];
        if (I != J)
            for (int i = 0; i < S; ++i)
                for (int j = 0; j < S; ++j) ke[(size_t)(S * I + i) * (S * N) + S * J + j] = q1[i][j];
        return;
    }
    const unsigned ni = (unsigned)lds_i[L.o_cn + u * N + I], nj = (unsigned)lds_i[L.o_cn + u * N + J];
    for (int side = 0; side < (I != J ? 2 : 1); ++side) {
        const int R = side ? J : I;
        const unsigned nr = side ? nj : ni, nc = side ? ni : nj;
        unsigned r0, cnt;
        int pos;
        if (nc_lds) {
            r0 = (unsigned)lds_i[L.o_ncr + 2 * (u * N + R)];
            cnt = (unsigned)lds_i[L.o_ncr + 2 * (u * N + R) + 1];
            pos = find_col_lds(lds_i + L.o_nc + (u * N + R) * a.nc_row, (int)cnt, (int)nc);
        } else {
            r0 = a.noff[nr];
            cnt = a.noff[nr + 1] - r0;
            pos = find_col(a.ncols + r0, (int)cnt, nc);
        }
        double* base = a.vals + (size_t)S * S * r0 + (size_t)S * pos;
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j) {
                if (side == 0) add_value<MODE>(base + (size_t)i * S * cnt + j, p1[i][j]);
                else add_value<MODE>(base + (size_t)j * S * cnt + i, q1[i][j]);
            }
    }
}

// ============================================================================================ matrix
template <int EK, int OP, int MODE>
__global__ void __launch_bounds__(256) k_assemble_matrix(const KArgs a) {
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    constexpr int D = E::D, N = E::N, S = O::S;
    (void)D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool GATHER = (MODE == MODE_GATHER);
    const Layout L = make_layout<EK, OP, WHAT_MATRIX>(a.nq, a.ub, a.acc_max, a.nb_max, GATHER, a.mb, a.fast, 0, GATHER ? 0 : a.nc_row);
    double* lds = reinterpret_cast<double*>(smem);
    int* lds_i = reinterpret_cast<int*>(smem + sizeof(double) * (size_t)L.n_doubles);
    const int tid = threadIdx.x, nt = blockDim.x;

    stage_tables<EK>(a, L, lds);

    if (!GATHER) {
        // ---- element-centric unit: elements [e0, e0 + U) of the (colour) list
        const long long w0 = a.work_begin + (long long)blockIdx.x * a.epb;
        const int U = (int)min((long long)a.epb, a.work_end - w0);
        for (int i = tid; i < U; i += nt) lds_i[L.o_uniq + i] = a.labels ? (int)a.labels[w0 + i] : (int)(w0 + i);
        __syncthreads();
        stage_elements<EK, S>(a, L, lds, lds_i, U, O::NEEDS_U, lds_i + L.o_uniq);
        const bool nc_lds = (a.nc_row > 0) && (MODE != MODE_DUMP);
        if (nc_lds) {
            for (int i = tid; i < U * N; i += nt) {
                const unsigned node = (unsigned)lds_i[L.o_cn + i];
                const unsigned r0 = a.noff[node];
                lds_i[L.o_ncr + 2 * i] = (int)r0;
                lds_i[L.o_ncr + 2 * i + 1] = (int)(a.noff[node + 1] - r0);
            }
            __syncthreads();
            for (int i = tid; i < U * N * a.nc_row; i += nt) {
                const int un = i / a.nc_row, k = i % a.nc_row;
                if (k < lds_i[L.o_ncr + 2 * un + 1]) lds_i[L.o_nc + i] = (int)a.ncols[(unsigned)lds_i[L.o_ncr + 2 * un] + k];
            }
        }
        for (int i = tid; i < U * a.nq; i += nt)
            prologue<EK, OP, WHAT_MATRIX>(a, L, lds, lds_i, i / a.nq, i % a.nq, lds_i + L.o_uniq + i / a.nq);
        __syncthreads();
        constexpr int NP = N * (N + 1) / 2;
        for (int it = tid; it < U * NP; it += nt) {
            const int u = it / NP;
            int I, J;
            unpack_pair(it % NP, I, J);
            double blk[S][S];
            pair_block<EK, OP>(a, L, lds, a.nq, u, I, J, blk);
            if constexpr (OP == FH_TENSOR) {
                // Symmetry::NonSymmetric (operators.rs:178-181): K_JI is a block of its own, not the mirror image of K_IJ.  Below, `blk` is what
                // goes to (I, J) and its transpose to (J, I): with P = K_IJ and Q = K_JI that is blk = P for the first and blk = Q^T for the
                // second store -- or, for the transposed element matrices of the two-pass assembly (nonsym bit 1), Q^T and P.
                if (a.nonsym & 1) {
                    double b2[S][S], p1[S][S], q1[S][S];
                    if (I != J) pair_block<EK, OP>(a, L, lds, a.nq, u, J, I, b2);
#pragma unroll
                    for (int i = 0; i < S; ++i)
#pragma unroll
                        for (int j = 0; j < S; ++j) {
                            const double kji_t = (I != J) ? b2[j][i] : blk[j][i];   // (K_JI)^T [i][j]
                            p1[i][j] = (a.nonsym & 2) ? kji_t : blk[i][j];
                            q1[i][j] = (a.nonsym & 2) ? blk[i][j] : kji_t;
                        }
                    tensor_store<EK, OP, MODE>(a, L, lds_i, nc_lds, u, I, J, w0, p1, q1);
                    continue;
                }
            }
            if (MODE == MODE_DUMP && S == 3 && a.ke_tri) {
                // the lower node-block triangle, row-major: block (J, I), I <= J, at place J (J + 1) / 2 + I -- the pair index this thread unpacked, so
                // consecutive threads write consecutive runs of 72 bytes and nothing is written twice (the full column-major matrix below leaves in
                // 24-byte pieces, both triangles: the pass was bound by those stores).  K[(J, r), (I, c)] = K[(I, c), (J, r)] = blk[c][r].
                typedef double f64x2_u8 __attribute__((ext_vector_type(2), aligned(8)));
                double* kb = a.ke_out + ((size_t)(a.ke_by_elem ? (long long)lds_i[L.o_uniq + u] : (w0 - a.work_begin + u)) * NP + (size_t)(it % NP)) * 9;
                const double v[9] = {blk[0][0], blk[1 % S][0], blk[2 % S][0], blk[0][1 % S], blk[1 % S][1 % S], blk[2 % S][1 % S],
                                     blk[0][2 % S], blk[1 % S][2 % S], blk[2 % S][2 % S]};
#pragma unroll
                for (int k = 0; k < 4; ++k) { f64x2_u8 p2; p2.x = v[2 * k]; p2.y = v[2 * k + 1]; reinterpret_cast<f64x2_u8*>(kb)[k] = p2; }
                kb[8] = v[8];
            } else if (MODE == MODE_DUMP) {
                // K_e column-major (s n) x (s n), both triangles
                double* ke = a.ke_out + (size_t)(a.ke_by_elem ? (long long)lds_i[L.o_uniq + u] : (w0 - a.work_begin + u)) * (S * N) * (S * N);
                // runs of S contiguous values: column (J, j) rows (I, 0..S), and column (I, i) rows (J, 0..S)
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    double* col = ke + (size_t)(S * J + j) * (S * N) + S * I;
                    if (S == 3) {
                        typedef double f64x2_u8 __attribute__((ext_vector_type(2), aligned(8)));
                        f64x2_u8 p2; p2.x = blk[0][j]; p2.y = blk[1 % S][j];
                        *reinterpret_cast<f64x2_u8*>(col) = p2;
                        col[2 % S] = blk[2 % S][j];
                    } else {
#pragma unroll
                        for (int i = 0; i < S; ++i) col[i] = blk[i][j];
                    }
                }
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    double* col = ke + (size_t)(S * I + i) * (S * N) + S * J;
                    if (S == 3) {
                        typedef double f64x2_u8 __attribute__((ext_vector_type(2), aligned(8)));
                        f64x2_u8 p2; p2.x = blk[i][0]; p2.y = blk[i][1 % S];
                        *reinterpret_cast<f64x2_u8*>(col) = p2;
                        col[2 % S] = blk[i][2 % S];
                    } else {
#pragma unroll
                        for (int j = 0; j < S; ++j) col[j] = blk[i][j];
                    }
                }
            } else {
                const unsigned ni = (unsigned)lds_i[L.o_cn + u * N + I], nj = (unsigned)lds_i[L.o_cn + u * N + J];
                {
                    unsigned r0, cnt;
                    int pos;
                    if (nc_lds) {
                        r0 = (unsigned)lds_i[L.o_ncr + 2 * (u * N + I)];
                        cnt = (unsigned)lds_i[L.o_ncr + 2 * (u * N + I) + 1];
                        pos = find_col_lds(lds_i + L.o_nc + (u * N + I) * a.nc_row, (int)cnt, (int)nj);
                    } else {
                        r0 = a.noff[ni];
                        cnt = a.noff[ni + 1] - r0;
                        pos = find_col(a.ncols + r0, (int)cnt, nj);
                    }
                    double* base = a.vals + (size_t)S * S * r0 + (size_t)S * pos;
#pragma unroll
                    for (int i = 0; i < S; ++i)
#pragma unroll
                        for (int j = 0; j < S; ++j) add_value<MODE>(base + (size_t)i * S * cnt + j, blk[i][j]);
                }
                if (I != J) {
                    unsigned r0, cnt;
                    int pos;
                    if (nc_lds) {
                        r0 = (unsigned)lds_i[L.o_ncr + 2 * (u * N + J)];
                        cnt = (unsigned)lds_i[L.o_ncr + 2 * (u * N + J) + 1];
                        pos = find_col_lds(lds_i + L.o_nc + (u * N + J) * a.nc_row, (int)cnt, (int)ni);
                    } else {
                        r0 = a.noff[nj];
                        cnt = a.noff[nj + 1] - r0;
                        pos = find_col(a.ncols + r0, (int)cnt, ni);
                    }
                    double* base = a.vals + (size_t)S * S * r0 + (size_t)S * pos;
#pragma unroll
                    for (int j = 0; j < S; ++j)
#pragma unroll
                        for (int i = 0; i < S; ++i) add_value<MODE>(base + (size_t)j * S * cnt + i, blk[i][j]);
                }
            }
        }
        return;
    }

    // ---- owner-computes unit: nodes [i0, i0 + nb) described by the precomputed block tables
    const GatherHdr hdr = a.gt_hdr[blockIdx.x];
    const int i0 = hdr.i0, nb = hdr.nb, r0 = hdr.r0, nrow = hdr.nrow, k0 = hdr.k0, m = hdr.m, U = hdr.U;
    const int nacc = S * S * nrow;
    double* acc = lds + L.o_ACC;
    for (int i = tid; i <= nb; i += nt) lds_i[L.o_noff + i] = (int)a.noff[i0 + i];
    if (!a.gt_pos)
        for (int i = tid; i < nrow; i += nt) lds_i[L.o_ncols + i] = (int)a.ncols[r0 + i];
    const bool ent_in_lds = m <= a.mb;  // U <= m, so the unique list fits whenever the entries do
    if (ent_in_lds)
        for (int i = tid; i < U; i += nt) lds_i[L.o_uniq + i] = (int)a.gt_elems[hdr.u_off + i];
    const int* uniq = ent_in_lds ? lds_i + L.o_uniq : reinterpret_cast<const int*>(a.gt_elems + hdr.u_off);
    unsigned char* pos_lds = reinterpret_cast<unsigned char*>(lds_i + L.o_pos);
    if (ent_in_lds) {
        for (int i = tid; i < m; i += nt) lds_i[L.o_ent + i] = (int)a.gt_ent[k0 + i];
        if (a.gt_pos)   // (null when a row has 256 or more column blocks: the byte-wide slots do not exist, the columns are searched)
            for (int i = tid; i < m * N; i += nt) pos_lds[i] = a.gt_pos[(size_t)k0 * N + i];
    }
    for (int i = tid; i < nacc; i += nt) acc[i] = 0.0;
    __syncthreads();

    for (int c0 = 0; c0 < U; c0 += a.ub) {
        const int Uc = min(a.ub, U - c0);
        stage_elements<EK, S>(a, L, lds, lds_i, Uc, O::NEEDS_U, uniq + c0);
        // phase B
        for (int i = tid; i < Uc * a.nq; i += nt)
            prologue<EK, OP, WHAT_MATRIX>(a, L, lds, lds_i, i / a.nq, i % a.nq, uniq + c0 + i / a.nq);
        __syncthreads();
        // phase C: one lane per (entry, other local node); entry = (unique slot, local index a, local node)
        for (int it = tid; it < m * N; it += nt) {
            const int t = it / N, Jn = it % N;
            const unsigned packed = ent_in_lds ? (unsigned)lds_i[L.o_ent + t] : a.gt_ent[k0 + t];
            const int u = (int)(packed >> 16) - c0;
            if (u < 0 || u >= Uc) continue;
            const int an = (int)((packed >> 8) & 0xffu);
            const int il = (int)(packed & 0xffu);
            double blk[S][S];
            const bool swap = an > Jn && !(OP == FH_TENSOR && (a.nonsym & 1));   // (NonSymmetric: the block of the ordered pair itself)
            pair_block<EK, OP>(a, L, lds, a.nq, u, swap ? Jn : an, swap ? an : Jn, blk);
            const int rb = lds_i[L.o_noff + il] - r0, cnt = lds_i[L.o_noff + il + 1] - lds_i[L.o_noff + il];
            int pos;
            if (a.gt_pos) {
                pos = ent_in_lds ? (int)pos_lds[it] : (int)a.gt_pos[(size_t)k0 * N + it];
            } else {
                pos = find_col_lds(lds_i + L.o_ncols + rb, cnt, lds_i[L.o_cn + u * N + Jn]);
            }
            double* base = acc + S * S * rb + S * pos;
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int j = 0; j < S; ++j) atomic_add_f64(base + i * S * cnt + j, swap ? blk[j][i] : blk[i][j]);
        }
        __syncthreads();
    }
    // phase D: rows [S i0, S (i0 + nb)) are contiguous in the CSR values
    double* out = a.vals + (size_t)S * S * r0;
    if (a.overwrite) for (int i = tid; i < nacc; i += nt) out[i] = acc[i];
    else for (int i = tid; i < nacc; i += nt) out[i] += acc[i];
}

// ============================================================================================ pipelined gather
// Persistent owner-computes kernel: each workgroup walks blocks b = blockIdx.x, += gridDim.x and keeps the
// global-memory latency off the critical path with a two-stage register prefetch:
//   top of iteration b:   issue vertex loads of block b+G (its geometry-node indices arrived last iteration)
//                         issue record loads of block b+2G (fixed-stride tables: address depends on b only)
//   body:                 phases B, C, D of block b from LDS
//   bottom:               park the prefetched block b+G in LDS (X, entries, column slots, row offsets)
// Only for operators that do not read u (Laplace, LinearElastic) and elements with few geometry nodes.
struct PipeTables {
    // all tables are indexed by POSITION p in the sweep order (blocks reordered into chains whose consecutive
    // members share elements; the shared elements stay staged in LDS from one block to the next)
    // rec: one packed record of rw 32-bit words per position, laid out exactly like its LDS copy:
    //   [0, 8)            GatherHdr of the block; k0 holds the number of NEW slots
    //   [8, +us/4)        occupied slots, 1 byte each: the new ones first, then the retained
    //   [.., +ms)         packed entries (persistent slot << 16 | a << 8 | local node)
    //   [.., +ms*N/4)     column slots, 4 per word
    //   [.., +nbs+1)      node-level row offsets relative to the block start
    const int* rec;             // [npos][rw]
    const int* conn;            // [npos][cs]   geometry-node indices per LDS slot (padded with 0)
    const int* elem;            // [npos][us]   element id per slot (error reporting)
    const double* slotpar;      // [npos][us][2] (mu, lambda) of the element in each slot -- compact table whose rules are
                                //              constant over the points (piecewise-constant material); else null
    int rw;                     // words per record
    int cs, ms, nbs, us;        // strides: cs = us * NG
    int npos;
};

__host__ __device__ inline int pipe_record_words(int us, int ms, int N, int nbs) { return 8 + us / 4 + ms + ms * N / 4 + nbs + 1; }

// DBG: the profiling hooks (FENRIS_HIP_TRACE phase stamps, FENRIS_HIP_ABLATE switches) exist only in the DBG instantiation;
// the production kernel carries none of their scalar branches.
// FULLQ: the rule has exactly QC points (one chunk): the chunk loop and its bounds are compile-time; the host also
// guarantees cs <= 256 and rw <= 256, so one geometry-node slot and one record word per thread suffice.
// ELEMPAR: (mu, lambda) per element (T.slotpar) instead of one uniform pair: the gradients are still stored pre-scaled
// by sqrt(w |det J|), the parameters only enter the finalize of a lane, whose two blocks belong to one element.
template <int EK, int OP, int QC, int JT, bool DBG = false, bool FULLQ = false, bool ELEMPAR = false>
__global__ void __launch_bounds__(256, (QC >= 8 ? 2 : 3)) k_gather_pipelined(const KArgs a, const PipeTables T) {
    const int ablate = DBG ? a.ablate : 0;
    const int nq_rt = FULLQ ? QC : a.nq;
    unsigned long long* const trace = DBG ? a.trace : nullptr;
    // JT = local nodes J handled per lane in phase C
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    constexpr int D = E::D, N = E::N, NG = E::NG, S = O::S;
    constexpr int SLOTS = FULLQ ? 1 : 2;  // geometry-node slots per thread: U * NG <= 512 (<= 256 with FULLQ)
    // PLANAR: component-major gradient rows, phase C fetches the two column nodes of a lane with one ds_read_b128
    // per component (pipelined_planar() in engine.hip makes the same choice for the LDS size)
    constexpr bool PLANAR = FULLQ && N == 8 && D == 3 && JT == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout L = make_layout<EK, OP, WHAT_MATRIX>(a.nq, a.ub, a.acc_max, a.nb_max, true, a.mb, 1, QC, 0, PLANAR ? 1 : 0);
    double* lds = reinterpret_cast<double*>(smem);
    int* lds_i = reinterpret_cast<int*>(smem + sizeof(double) * (size_t)L.n_doubles);
    double* acc = lds + L.o_ACC;
    const int tid = threadIdx.x, nt = 256;
    const int G = gridDim.x;
    stage_tables<EK>(a, L, lds);

    // Prefetch state per thread: one header word (threads 0..7), SLOTS geometry-node indices, one packed entry,
    // one word of column slots, one relative row offset.  Headers are parked in LDS (double-buffered by block
    // parity) and read back as wave-uniform values, so they cost one VGPR instead of eight.
    constexpr int RL = FULLQ ? 1 : 2;  // record words per thread: rw <= 512 (checked by the host; <= 256 with FULLQ)
    struct Rec { int w[RL]; int conn[SLOTS]; };
    const int npos = T.npos;
    // contiguous range of positions per workgroup: consecutive positions are chain neighbours
    const int p_begin = (int)((long long)blockIdx.x * npos / G), p_end = (int)((long long)(blockIdx.x + 1) * npos / G);
    // every lane loads (clamped index, same cache lines) so that the prefetch is branch-free: with predicated
    // loads the compiler's waitcnt bookkeeping merges divergent paths and falls back to vmcnt(0) right after
    // the issue, which serialises the prefetch with the block's compute
    auto load_rec = [&](int p, Rec& r) {
        p = min(p, npos - 1);
#pragma unroll
        for (int k = 0; k < RL; ++k) r.w[k] = T.rec[(size_t)p * T.rw + min(tid + k * nt, T.rw - 1)];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) r.conn[k] = T.conn[(size_t)p * T.cs + min(tid + k * nt, T.cs - 1)];
    };
    // pins the record in registers: the compiler places the vmcnt wait for its loads here
    auto land_rec = [&](Rec& r) {
        if constexpr (FULLQ) asm volatile("" : "+v"(r.w[0]), "+v"(r.conn[0]));
        else asm volatile("" : "+v"(r.w[0]), "+v"(r.w[RL - 1]), "+v"(r.conn[0]), "+v"(r.conn[SLOTS - 1]));
    };
    double V[SLOTS][D];
    auto load_verts = [&](const Rec& r) {
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
#pragma unroll
            for (int c = 0; c < D; ++c) V[k][c] = a.verts[(size_t)((ablate & 512) ? (r.conn[k] & 63) : r.conn[k]) * D + c];
    };
    // integer LDS: one packed record (see PipeTables) per block parity.  Everything parked for block p+1 is
    // double-buffered, so parking needs no barrier of its own; X is single-buffered: it is only read by phase B,
    // which lies behind the barrier that precedes phase C.
    auto rec_base = [&](int parity) { return lds_i + parity * T.rw; };
    auto park = [&](const Rec& r, int parity) {  // registers -> LDS for the block that is computed next
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int sidx = tid + k * nt;
            if (sidx < T.cs)  // padded / retained slots carry valid vertex indices: harmless
#pragma unroll
                for (int c = 0; c < D; ++c) lds[L.o_X + (sidx / NG) * L.xs + (sidx % NG) * D + c] = V[k][c];
        }
        int* rb = rec_base(parity);
#pragma unroll
        for (int k = 0; k < RL; ++k)
            if (tid + k * nt < T.rw) rb[tid + k * nt] = r.w[k];
    };
    // phase D: rows of a finished block -> global memory, accumulators cleared.  NTH threads take part, in
    // reverse thread order, so that the waves with no phase-B work do it while the others run phase B.
    auto write_out = [&](double* out, int nacc, int nth) {
        const int rt = nt - 1 - tid;
        if (rt >= nth) return;
        if (ablate & 8) {
            for (int i = rt; i < nacc; i += nth) acc[i] = 0.0;
        } else if (PLANAR && a.overwrite) {
            // pairs: ds_read_b128 / global_store_dwordx4 / ds_write_b128 (the accumulators start 16-byte aligned;
            // the rows in global memory are 8-byte aligned, which the vector-memory path accepts)
            typedef double f64x2_u __attribute__((ext_vector_type(2), aligned(8)));
            f64x2* acc2 = reinterpret_cast<f64x2*>(acc);
            f64x2_u* out2 = reinterpret_cast<f64x2_u*>(out);
            const int npair = nacc >> 1;
            const f64x2 zero2 = {0.0, 0.0};
            int i = rt;
            for (; i + nth < npair; i += 2 * nth) {
                const f64x2 v0 = acc2[i], v1 = acc2[i + nth];
                out2[i] = v0; out2[i + nth] = v1;
                acc2[i] = zero2; acc2[i + nth] = zero2;
            }
            for (; i < npair; i += nth) { out2[i] = acc2[i]; acc2[i] = zero2; }
            if ((nacc & 1) && rt == 0) { out[nacc - 1] = acc[nacc - 1]; acc[nacc - 1] = 0.0; }
        } else if (a.overwrite) {
            int i = rt;
            for (; i + 3 * nth < nacc; i += 4 * nth) {  // batches of four: the LDS reads overlap
                const double v0 = acc[i], v1 = acc[i + nth], v2 = acc[i + 2 * nth], v3 = acc[i + 3 * nth];
                out[i] = v0; out[i + nth] = v1; out[i + 2 * nth] = v2; out[i + 3 * nth] = v3;
                acc[i] = 0.0; acc[i + nth] = 0.0; acc[i + 2 * nth] = 0.0; acc[i + 3 * nth] = 0.0;
            }
            for (; i < nacc; i += nth) { out[i] = acc[i]; acc[i] = 0.0; }
        } else {
            for (int i = rt; i < nacc; i += nth) { out[i] += acc[i]; acc[i] = 0.0; }
        }
    };

    // FULLQ: a lane always works on point tid % QC in phase B
    const double sqw = PLANAR ? sqrt(a.qw[tid % QC]) : 0.0;
    int p = p_begin;
    if (p >= p_end) return;
    Rec nxt;
    {
        Rec cur;
        load_rec(p, cur);
        load_verts(cur);
        load_rec(p + 1, nxt);
        park(cur, 0);
        land_rec(nxt);
    }
    for (int i = tid; i < a.acc_max; i += nt) acc[i] = 0.0;
    __syncthreads();

    // optional phase timing (FENRIS_HIP_TRACE): lane 0 of every wave accumulates s_memtime deltas per phase
    unsigned long long tr_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tr_t = trace ? __builtin_amdgcn_s_memtime() : 0;
#define FH_STAMP(K)                                              \
    if (trace) {                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        tr_acc[K] += now_ - tr_t;                                \
        tr_t = now_;                                             \
    }
    int parity = 0;
    double* prev_out = nullptr;  // rows of the previous block, still in the accumulators
    int prev_nacc = 0;
    for (; p < p_end; ++p, parity ^= 1) {
        const bool have_next = (p + 1) < p_end;
        // LDS reads of this block's header and item first, the global prefetch is issued while they are in flight
        const int* rec = rec_base(parity);
        const GatherHdr hc = *reinterpret_cast<const GatherHdr*>(rec);
        const unsigned char* slot_b = reinterpret_cast<const unsigned char*>(rec + 8);
        const int* ent_l = rec + 8 + T.us / 4;
        const unsigned char* pos_b = reinterpret_cast<const unsigned char*>(rec + 8 + T.us / 4 + T.ms);
        const int* noff_l = rec + 8 + T.us / 4 + T.ms + T.ms * N / 4;
        constexpr int NGRP = (JT <= N) ? N / JT : 1;
        // items are sorted by node: consecutive items accumulate into the same rows, often into the same values (the
        // diagonal block of a node receives one contribution per adjacent element: 24 on a BCC tetrahedral mesh), and
        // ds_add_f64 serialises lanes of one instruction that hit one address.  Deal the items round-robin to the four
        // waves, so that one instruction sees a quarter of a node's contributions.  Simplices only (C3: 1.45 -> 1.31 ms):
        // a hexahedral node has 8 elements, and there the locality of consecutive items is worth more (+4.5 % when dealt).
        constexpr int IPW = (NGRP <= 64) ? 64 / NGRP : 1;  // items per wave
        constexpr bool DEAL = (N == D + 1) && NGRP <= 64;
        const int g_item = tid / NGRP;
        const int t_item = DEAL ? (g_item % IPW) * 4 + g_item / IPW : g_item;
        const int j0 = (tid % NGRP) * JT;
        const unsigned packed_raw = (unsigned)ent_l[min(t_item, T.ms - 1)];
        load_verts(nxt);      // lands while this block is computed (clamped past the end: harmless)
        Rec nn;
        load_rec(p + 2, nn);
        // slots to (re)compute: only the new ones, except at the start of this workgroup's range where nothing
        // is staged yet (the slot list holds the new slots first, then the retained ones)
        const int U = (p == p_begin || QC < nq_rt) ? hc.U : hc.k0;  // chunked staging keeps nothing across blocks
        const int m = hc.m, nrow = hc.nrow;
        const int nacc = S * S * nrow;
        // G = sum_q h_I h_J^T accumulated in registers across chunks of QC quadrature points; all unique
        // elements of the block are staged at once (U <= ub guaranteed by the host).  One lane owns an entry
        // (node, element, local index a) and JT consecutive local nodes J: h_a is read once per point for JT
        // blocks (LDS traffic, not VALU, bounds this kernel).
        double Gr[JT][D][D];
#pragma unroll
        for (int r = 0; r < JT; ++r)
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) Gr[r][i][j] = 0.0;
        const bool has_item = t_item < m;
        const unsigned packed = has_item ? packed_raw : 0u;
        const int u_item = (int)(packed >> 16);
        // material of this lane's element: fetched now, used by the finalize a whole block later
        double mu_i = a.mu, lambda_i = a.lambda;
        if constexpr (ELEMPAR) {
            const double* sp = T.slotpar + 2 * ((size_t)p * T.us + u_item);
            mu_i = sp[0];
            lambda_i = sp[1];
        }
        const int an = (int)((packed >> 8) & 0xffu);
        const int il = (int)(packed & 0xffu);
        for (int qc = 0; qc < nq_rt; qc += QC) {
            if (qc > 0) lds_barrier();  // the previous chunk's phase C is done with the staged points
            FH_STAMP(0)  // top of block: prefetch issue, header, item decode
            // phase B for quadrature points [qc, qc + QC)
            if constexpr (FULLQ) {  // U <= 32 slots x 8 points: at most one item per thread
                if (tid < U * QC && !(ablate & 1)) {
                    const int u = (int)slot_b[tid / QC];
                    prologue<EK, OP, WHAT_MATRIX, PLANAR>(a, L, lds, lds_i, u, tid % QC, T.elem + (size_t)p * T.us + u, tid % QC, sqw);
                }
            } else if (!(ablate & 1)) {
                for (int i = tid; i < U * QC; i += nt) {
                    const int u = (int)slot_b[i / QC], qs = i % QC;
                    if (qc + qs < nq_rt)
                        prologue<EK, OP, WHAT_MATRIX>(a, L, lds, lds_i, u, qc + qs, T.elem + (size_t)p * T.us + u, qs);
                }
            }
            // phase D of the previous block, overlapped with phase B: its accumulators are complete (barrier at the
            // end of the last iteration) and are not touched again before the barrier below
            // (phase B fills half of the waves: the other half writes out; a smaller phase B is not worth idling for)
            if (qc == 0 && prev_nacc > 0 && !(ablate & 16))
                write_out(prev_out, prev_nacc, (U * QC > nt / 4 && U * QC <= nt / 2) ? nt / 2 : nt);
            FH_STAMP(1)  // phase B (+ write-out of the previous block)
            lds_barrier();
            FH_STAMP(2)  // barrier after B
            // phase C (accumulate)
            const int nqc = FULLQ ? QC : min(QC, nq_rt - qc);
            if constexpr (PLANAR) {
                if (has_item && !(ablate & 2)) {
                    // straight-line, all strides compile-time: two address registers and immediate offsets.  Per
                    // point 3 ds_read_b64 (row node, one per component) + 3 ds_read_b128 (the two column nodes);
                    // two points (12 operations, the lgkm counter tracks 15) are in flight while one is multiplied.
                    constexpr int QPD = N * D + 2, AHEAD = 2, NB = AHEAD + 1;
                    const double* pq = lds + L.o_QP + (size_t)u_item * L.qss;
                    const unsigned pa = (unsigned)(unsigned long long)(pq + an), pb = (unsigned)(unsigned long long)(pq + j0);
                    double av[NB][D];
                    f64x2 bv[NB][D];
                    auto fetchq = [&](auto qk) {
                        constexpr int qq = decltype(qk)::value, sl = qq % NB;
                        av[sl][0] = lds_read_f64_at<(qq * QPD) * 8>(pa);
                        av[sl][1] = lds_read_f64_at<(qq * QPD + N) * 8>(pa);
                        av[sl][2] = lds_read_f64_at<(qq * QPD + 2 * N) * 8>(pa);
                        bv[sl][0] = lds_read_f64x2<(qq * QPD) * 8>(pb);
                        bv[sl][1] = lds_read_f64x2<(qq * QPD + N) * 8>(pb);
                        bv[sl][2] = lds_read_f64x2<(qq * QPD + 2 * N) * 8>(pb);
                    };
                    fetchq(std::integral_constant<int, 0>{});
                    fetchq(std::integral_constant<int, 1>{});
                    pipeline_consume<QC, D>([&](auto qk) {
                        constexpr int qq = decltype(qk)::value, sl = qq % NB;
                        // wait for point qq first (point qq + 1 may stay in flight), then issue point qq + AHEAD: never
                        // more than 12 operations outstanding
                        constexpr int ahead = (QC - 1 - qq) < (AHEAD - 1) ? (QC - 1 - qq) : (AHEAD - 1);
                        lds_wait<ahead * 2 * D>();
                        if constexpr (qq + AHEAD < QC) fetchq(std::integral_constant<int, qq + AHEAD>{});
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < D; ++i)
#pragma unroll
                            for (int j = 0; j < D; ++j) {
                                Gr[0][i][j] = fma(av[sl][i], bv[sl][j].x, Gr[0][i][j]);
                                Gr[1][i][j] = fma(av[sl][i], bv[sl][j].y, Gr[1][i][j]);
                            }
                    });
                }
            } else if (has_item && !(ablate & 2)) {
                const double* pq = lds + L.o_QP + (size_t)u_item * L.qss;
                // every fetch is an explicit ds_read_b64 (2 LDS cycles per wave): hipcc would merge neighbours
                // into ds_read2_b64, which costs 8 cycles for the same 16 bytes (MI355X_MICROARCH.md, LDS
                // table), and the LDS pipe -- shared by the CU's four SIMDs -- is what bounds this loop.
                // asm loads are invisible to the compiler's waitcnt insertion: explicit waits + sched_barrier.
                // Software pipeline in straight-line groups of GQ points: the fetches of point k+1 are in
                // flight while the products of point k issue (no in-flight value crosses a loop back-edge).
                constexpr int NL = D * (1 + JT);  // fetches per point
                auto fetch = [&](const double* pp, double (&av)[D], double (&bvv)[JT][D]) {
                    lds_read_vec<D>(pp + an * D, av);
#pragma unroll
                    for (int r = 0; r < JT; ++r) lds_read_vec<D>(pp + (j0 + r) * D, bvv[r]);
                };
                auto products = [&](const double (&av)[D], const double (&bvv)[JT][D]) {
                    // G[r] = h_a h_J^T; the (min, max)-role block is G or its transpose (same products)
#pragma unroll
                    for (int r = 0; r < JT; ++r)
#pragma unroll
                        for (int i = 0; i < D; ++i)
#pragma unroll
                            for (int j = 0; j < D; ++j) Gr[r][i][j] = fma(av[i], bvv[r][j], Gr[r][i][j]);
                };
                constexpr int GQ = (QC >= 4 && JT <= 2) ? 4 : 1;
                int q = 0;
                if (GQ > 1) {
                    for (; q + GQ <= nqc; q += GQ, pq += GQ * L.qpd) {
                        double av[2][D], bvv[2][JT][D];
                        fetch(pq, av[0], bvv[0]);
#pragma unroll
                        for (int k = 0; k < GQ; ++k) {
                            if (k + 1 < GQ) {
                                fetch(pq + (k + 1) * L.qpd, av[(k + 1) & 1], bvv[(k + 1) & 1]);
                                lds_wait<NL>();
                            } else {
                                lds_wait<0>();
                            }
                            products(av[k & 1], bvv[k & 1]);
                        }
                    }
                }
                for (; q < nqc; ++q, pq += L.qpd) {
                    double av[D], bvv[JT][D];
                    fetch(pq, av, bvv);
                    lds_wait_all();
                    products(av, bvv);
                }
            }
        }
        FH_STAMP(3)  // phase C
        // finalize: s x s block from G, mirrored like clone_upper_to_lower, then row accumulators
        if (has_item && !(ablate & 4)) {
            const int rb = noff_l[il], cnt = noff_l[il + 1] - rb;
#pragma unroll
            for (int r = 0; r < JT; ++r) {
                const int Jn = j0 + r;
                // value for (row comp i of node a, col comp j of node J): the block of the ordered pair
                // (min, max) is  mu (tr I + Gm^T) + lambda Gm  with Gm = G (a < J) or G^T (a > J); read through
                // the transpose when a > J, so  out[i][j] = mu (tr d_ij + G[i][j]... ) is written directly:
                //   a <= J: out[i][j] = mu (tr d_ij + G[j][i]) + lambda G[i][j]
                //   a >  J: out[i][j] = blk_(J,a)[j][i] = mu (tr d_ij + Gm[i][j]) + lambda Gm[j][i], Gm = G^T
                //                     = mu (tr d_ij + G[j][i]) + lambda G[i][j]      (same expression)
                double tr = 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) tr += Gr[r][i][i];
                const int pos = (int)pos_b[t_item * N + Jn];
                double* base = acc + S * S * rb + S * pos;
                if (OP == FH_LAPLACE) {
                    atomic_add_f64(base, tr);
                } else {
                    double val[D][D];
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j)
                            val[i][j] = (i == j) ? fma(mu_i, tr + Gr[r][i][i], lambda_i * Gr[r][i][i])
                                                 : fma(mu_i, Gr[r][j][i], lambda_i * Gr[r][i][j]);
                    // diagonal block (an == Jn): the lower triangle mirrors the upper one (util.rs:46-50).  Both
                    // operand vectors are then the same LDS values, so G[i][j] and G[j][i] are the same products summed
                    // in the same order -- val is symmetric bit for bit and needs no explicit mirroring.
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            const double v = val[i][j];
                            // experiments (wrong sums): 64 = lane-private, conflict-free targets; 32 = plain stores
                            double* dst = (ablate & 64) ? lds + L.o_QP + tid * 17 + r * 9 + i * 3 + j : base + (i % S) * S * cnt + (j % S);
                            if (ablate & 32) *dst = v;
                            else atomic_add_f64(dst, v);
                        }
                }
            }
        }
        // park the prefetched block (double-buffered regions, see above).  vmcnt counts loads and stores in one
        // queue: waiting for the prefetch here, a whole accumulate phase after the write-out stores were issued,
        // costs nothing.
        if (have_next) park(nxt, parity ^ 1);
        land_rec(nn);
        nxt = nn;
        prev_out = a.vals + (size_t)S * S * hc.r0;
        prev_nacc = nacc;
        FH_STAMP(4)  // finalize + park
        lds_barrier();
        if (ablate & 16) {  // experiment: write-out by all threads behind the barrier instead of overlapped
            write_out(prev_out, prev_nacc, nt);
            prev_nacc = 0;
            lds_barrier();
        }
        FH_STAMP(5)  // end barrier
    }
    write_out(prev_out, prev_nacc, nt);
#undef FH_STAMP
    if (trace && (tid & 63) == 0) {  // one row of 7 counters per wave index
        unsigned long long* row = trace + 7 * (tid >> 6);
        for (int k = 0; k < 6; ++k) atomicAdd(row + k, tr_acc[k]);
        atomicAdd(row + 6, 1ull);
    }
}

// (mu, lambda) of the element staged in every slot of every position, for compact tables whose rules are constant over
// their points: out[2 i] = rparams[rule_map[elem[i]]][point 0]
static __global__ void __launch_bounds__(256) k_build_slot_params(const int* elem, size_t n, const unsigned* rule_map, const double* rparams,
                                                           int nq, double* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = elem[i];
    double mu = 0.0, lambda = 0.0;
    if (e >= 0) {
        const double* par = rparams + (size_t)rule_map[e] * nq * 2;
        mu = par[0];
        lambda = par[1];
    }
    out[2 * i] = mu;
    out[2 * i + 1] = lambda;
}

// successor of a block in the sweep order: the block that owns most of the nodes (with an index beyond this
// block) touched by this block's elements -- i.e. the neighbouring block sharing the most elements.  One
// wavefront per block.
// cls (optional): class of every block; only blocks of the same class are candidates (chains never mix classes).
// r2v (optional): position of every node in the order the blocks were formed in (null: the node numbering itself).
template <int TBL>   // places of the hash table that counts the candidates: 512 (up to 256 candidates per block) or 2048 (up to 1024)
static __global__ void __launch_bounds__(64) k_block_successor(const GatherHdr* hdr, const unsigned* gt_elems, const int* conn, int N,
                                                        const int* node2blk, int nblk, const unsigned char* cls, int* succ,
                                                        const int* r2v = nullptr) {
    __shared__ int cand[1024];
    __shared__ int hkey[TBL], hcnt[TBL];
    __shared__ int best_blk, best_cnt;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (cls && cls[b] == 1) {   // an affine block: its kernel sweeps in CSR order, every position a chain of its own (build_partition)
        if (lane == 0) succ[b] = -1;
        return;
    }
    const GatherHdr h = hdr[b];
    const int last = h.i0 + h.nb - 1;
    const int total = min(h.U * N, 1024);
    for (int i = lane; i < total; i += 64) {
        const int node_id = conn[(size_t)gt_elems[h.u_off + i / N] * N + i % N];
        const int node = r2v ? r2v[node_id] : node_id;
        int cb = (node > last) ? node2blk[node] : -1;
        if (cls && cb >= 0 && cls[cb] != cls[b]) cb = -1;
        cand[i] = cb;
    }
    if (lane == 0) { best_blk = -1; best_cnt = 0; }
    if (2 * total <= TBL) {
        // round 5: the candidates counted in a hash table of 512 places (at most 256 distinct candidates: never full) instead of by comparing
        // every candidate with every other (2 x 256 x 256 LDS reads per block, 18.8 ms for the 1.46 M blocks of the 216^3 mesh); the result
        // -- largest count, smallest block among those -- does not depend on the order of the atomics
        for (int i = lane; i < TBL; i += 64) { hkey[i] = -1; hcnt[i] = 0; }
        __syncthreads();
        for (int i = lane; i < total; i += 64) {
            const int c = cand[i];
            if (c < 0) continue;
            unsigned hsh = ((unsigned)c * 2654435761u) >> (TBL == 512 ? 23 : 21);
            for (;;) {
                const int old = atomicCAS(&hkey[hsh], -1, c);
                if (old == -1 || old == c) { atomicAdd(&hcnt[hsh], 1); break; }
                hsh = (hsh + 1) & (unsigned)(TBL - 1);
            }
        }
        __syncthreads();
        int bc = 0, bb = 0x7fffffff;
        for (int i = lane; i < TBL; i += 64) {
            const int n = hcnt[i], k = hkey[i];
            if (n > bc || (n == bc && n > 0 && k < bb)) { bc = n; bb = k; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int oc = __shfl_xor(bc, o), ob = __shfl_xor(bb, o);
            if (oc > bc || (oc == bc && oc > 0 && ob < bb)) { bc = oc; bb = ob; }
        }
        if (lane == 0) succ[b] = (bc > 0) ? bb : -1;
        return;
    }
    __syncthreads();
    for (int i = lane; i < total; i += 64) {
        const int c = cand[i];
        if (c < 0) continue;
        int cnt = 0;
        bool first = true;
        for (int k = 0; k < total; ++k) {
            if (cand[k] == c) { ++cnt; if (k < i) first = false; }
        }
        if (first) atomicMax(&best_cnt, cnt);  // one representative per candidate
    }
    __syncthreads();
    for (int i = lane; i < total; i += 64) {
        const int c = cand[i];
        if (c < 0) continue;
        int cnt = 0;
        for (int k = 0; k < total; ++k) cnt += (cand[k] == c);
        if (cnt == best_cnt) atomicMin(reinterpret_cast<unsigned*>(&best_blk), (unsigned)c);  // -1 == 0xffffffff
    }
    __syncthreads();
    if (lane == 0) succ[b] = (best_cnt > 0) ? best_blk : -1;
    (void)nblk;
}

static __global__ void k_node_to_block(const unsigned* blk_off, int nblk, int* node2blk) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    for (unsigned i = blk_off[b]; i < blk_off[b + 1]; ++i) node2blk[i] = b;
}

// fixed-stride, position-indexed tables of the pipelined kernel.  One wavefront per CHAIN walks its positions in
// order and assigns every unique element of a block to an LDS slot: elements shared with the previous block of
// the chain keep their slot (their staged quadrature-point data is reused), new ones take the freed slots.
template <int NG_T>
__global__ void __launch_bounds__(64) k_build_pipe_tables(const int* order, const int* chain_off, const GatherHdr* hdr,
                                                          const unsigned* gt_elems, const unsigned* gt_ent,
                                                          const unsigned char* gt_pos, const unsigned* noff, const int* conn, int N,
                                                          int cs, int ms, int nbs, int us, int rw, int* p_rec, int* p_conn,
                                                          int* p_elem, const int by_parity) {
    __shared__ int slot_elem[256];   // element staged in each slot after the previous block (-1 = free)
    __shared__ int new_elem[256];    // element per slot after this block
    __shared__ int map[256];         // block-local unique index -> slot
    __shared__ unsigned char list[256];
    __shared__ unsigned char flist[256];   // free slots, ascending
    __shared__ unsigned Es[256];           // unique elements of the block
    __shared__ int hk[512];                // ... hashed: element id (-1: empty place)
    __shared__ unsigned char hv[512];      //             its index in the block's list
    __shared__ int s_nnew;
    const int lane = threadIdx.x;
    const int p0 = chain_off[blockIdx.x], p1 = chain_off[blockIdx.x + 1];
    for (int s_ = lane; s_ < us; s_ += 64) slot_elem[s_] = -1;
    __syncthreads();
    for (int p = p0; p < p1; ++p) {
        const GatherHdr h = hdr[order[p]];
        // (round 5: the block's element list in LDS -- the search below reads it us x U times; from global memory that was most of the 36 ms
        // this kernel took on the 5 M-element Tet4 mesh, ~180 elements per block)
        for (int k = lane; k < h.U && k < 256; k += 64) Es[k] = gt_elems[h.u_off + k];
        const unsigned* E = (h.U <= 256) ? Es : gt_elems + h.u_off;
        for (int s_ = lane; s_ < us; s_ += 64) new_elem[s_] = -1;
        for (int k = lane; k < h.U; k += 64) map[k] = -1;
        __syncthreads();
        // retained: an element of this block already staged in some slot keeps it.  (Round 5: the block's elements go into a hash table of 512
        // places -- at most 252 of them, all different -- and every staged slot looks its element up, instead of us x U comparisons.)
        if (h.U <= 256) {
            for (int i = lane; i < 512; i += 64) hk[i] = -1;
            __syncthreads();
            for (int k = lane; k < h.U; k += 64) {
                const int e = (int)E[k];
                unsigned hs = ((unsigned)e * 2654435761u) >> 23;
                while (atomicCAS(&hk[hs], -1, e) != -1) hs = (hs + 1) & 511u;
                hv[hs] = (unsigned char)k;
            }
            __syncthreads();
            for (int s_ = lane; s_ < us; s_ += 64) {
                const int e = slot_elem[s_];
                if (e < 0) continue;
                unsigned hs = ((unsigned)e * 2654435761u) >> 23;
                for (int key = hk[hs]; key != -1; hs = (hs + 1) & 511u, key = hk[hs])
                    if (key == e) { map[hv[hs]] = s_; new_elem[s_] = e; break; }
            }
        } else {
            for (int s_ = lane; s_ < us; s_ += 64) {
                const int e = slot_elem[s_];
                if (e < 0) continue;
                for (int k = 0; k < h.U; ++k)
                    if ((int)E[k] == e) { map[k] = s_; new_elem[s_] = e; break; }
            }
        }
        __syncthreads();
        if (!by_parity) {
            // new elements take the free slots in ascending order; list = new slots, then retained -- by the whole wavefront (ranks from
            // ballots): one lane walking U x us candidates was 0.36 s of set-up on the 5 M-element Tet4 mesh (180 elements per block)
            const unsigned long long below = (1ull << lane) - 1ull;
            int nfree = 0;
            for (int s0 = 0; s0 < us; s0 += 64) {
                const int s_ = s0 + lane;
                const bool fr = s_ < us && new_elem[s_] < 0;
                const unsigned long long m = __ballot(fr);
                if (fr) flist[nfree + __popcll(m & below)] = (unsigned char)s_;
                nfree += __popcll(m);
            }
            __syncthreads();
            int nnew = 0;
            for (int k0 = 0; k0 < h.U; k0 += 64) {
                const int k = k0 + lane;
                const bool nw = k < h.U && map[k] < 0;
                const unsigned long long m = __ballot(nw);
                const int r = nnew + __popcll(m & below);
                if (nw && r < nfree) {
                    const int sl = flist[r];
                    map[k] = sl;
                    new_elem[sl] = (int)E[k];
                    list[r] = (unsigned char)sl;
                }
                nnew += __popcll(m);
            }
            __syncthreads();
            int n = nnew;
            for (int s0 = 0; s0 < us; s0 += 64) {
                const int s_ = s0 + lane;
                const bool keep = s_ < us && new_elem[s_] >= 0 && slot_elem[s_] == new_elem[s_];
                const unsigned long long m = __ballot(keep);
                if (keep) list[n + __popcll(m & below)] = (unsigned char)s_;
                n += __popcll(m);
            }
            for (int i = n + lane; i < us; i += 64) list[i] = 0;
            if (lane == 0) s_nnew = nnew;
        } else {
            // by_parity (the general positions of Hex8 meshes: k_hex8_rows): a new element takes a free slot whose parity is that of the
            // element id when there is one -- that kernel keeps the gradients of even and odd slots in different halves of the LDS banks, and
            // on a structured mesh (an even number of cells per line) the elements that meet a node at the same local corner alternate in
            // parity.  (Not for the affine positions: their kernel's element records are 80 bytes apart, and the slot numbering this
            // gives them cost the headline 4 - 7 %.)
            // Round 5: by the whole wavefront, like the branch above -- the r-th new element of a parity takes the r-th free slot of that
            // parity; what is left over (a parity with more new elements than free slots) takes the remaining free slots in ascending order.
            // One lane walking U x us candidates was 22 ms of set-up on the 216^3 mesh.
            const unsigned long long below = (1ull << lane) - 1ull;
            int nfree2[2] = {0, 0};
            for (int s0 = 0; s0 < us; s0 += 64) {
                const int s_ = s0 + lane;
                const bool fr = s_ < us && new_elem[s_] < 0;
                const unsigned long long m0 = __ballot(fr && !(s_ & 1)), m1 = __ballot(fr && (s_ & 1));
                if (fr) flist[(s_ & 1) * 128 + nfree2[s_ & 1] + __popcll(((s_ & 1) ? m1 : m0) & below)] = (unsigned char)s_;
                nfree2[0] += __popcll(m0);
                nfree2[1] += __popcll(m1);
            }
            __syncthreads();
            int nn2[2] = {0, 0};
            for (int k0 = 0; k0 < h.U; k0 += 64) {
                const int k = k0 + lane;
                const bool nw = k < h.U && map[k] < 0;
                const int par = nw ? (int)(E[k] & 1u) : 0;
                const unsigned long long m0 = __ballot(nw && par == 0), m1 = __ballot(nw && par == 1);
                const int r = nn2[par] + __popcll((par ? m1 : m0) & below);
                if (nw && r < nfree2[par]) {
                    const int sl = flist[par * 128 + r];
                    map[k] = sl;
                    new_elem[sl] = (int)E[k];
                }
                nn2[0] += __popcll(m0);
                nn2[1] += __popcll(m1);
            }
            __syncthreads();
            // leftovers: the remaining free slots, ascending
            int nfree = 0;
            for (int s0 = 0; s0 < us; s0 += 64) {
                const int s_ = s0 + lane;
                const bool fr = s_ < us && new_elem[s_] < 0;
                const unsigned long long m = __ballot(fr);
                if (fr) flist[nfree + __popcll(m & below)] = (unsigned char)s_;
                nfree += __popcll(m);
            }
            __syncthreads();
            int nleft = 0;
            for (int k0 = 0; k0 < h.U; k0 += 64) {
                const int k = k0 + lane;
                const bool nw = k < h.U && map[k] < 0;
                const unsigned long long m = __ballot(nw);
                const int r = nleft + __popcll(m & below);
                if (nw && r < nfree) {
                    const int sl = flist[r];
                    map[k] = sl;
                    new_elem[sl] = (int)E[k];
                }
                nleft += __popcll(m);
            }
            __syncthreads();
            // list = the new slots in the order of their elements, then the retained ones
            int nnew = 0;
            for (int k0 = 0; k0 < h.U; k0 += 64) {
                const int k = k0 + lane;
                const int sl = (k < h.U) ? map[k] : -1;
                const bool nw = sl >= 0 && slot_elem[sl] != new_elem[sl];
                const unsigned long long m = __ballot(nw);
                if (nw) list[nnew + __popcll(m & below)] = (unsigned char)sl;
                nnew += __popcll(m);
            }
            int n = nnew;
            for (int s0 = 0; s0 < us; s0 += 64) {
                const int s_ = s0 + lane;
                const bool keep = s_ < us && new_elem[s_] >= 0 && slot_elem[s_] == new_elem[s_];
                const unsigned long long m = __ballot(keep);
                if (keep) list[n + __popcll(m & below)] = (unsigned char)s_;
                n += __popcll(m);
            }
            for (int i = n + lane; i < us; i += 64) list[i] = 0;
            if (lane == 0) s_nnew = nnew;
        }
        __syncthreads();
        // write the records of position p
        int* rec = p_rec + (size_t)p * rw;  // packed record, layout of PipeTables::rec
        unsigned* p_slots_r = reinterpret_cast<unsigned*>(rec + 8);
        unsigned* p_ent_r = reinterpret_cast<unsigned*>(rec + 8 + us / 4);
        unsigned* p_pos_r = reinterpret_cast<unsigned*>(rec + 8 + us / 4 + ms);
        int* p_noff_r = rec + 8 + us / 4 + ms + ms * N / 4;
        if (lane == 0) {
            GatherHdr o = h;
            o.k0 = s_nnew;
            *reinterpret_cast<GatherHdr*>(rec) = o;
        }
        for (int sidx = lane; sidx < cs; sidx += 64) {
            const int s_ = sidx / NG_T, g = sidx % NG_T;
            const int e = new_elem[s_];
            p_conn[(size_t)p * cs + sidx] = (e >= 0) ? conn[(size_t)e * N + g] : 0;
        }
        for (int s_ = lane; s_ < us; s_ += 64) p_elem[(size_t)p * us + s_] = new_elem[s_];
        for (int w = lane; w < us / 4; w += 64)
            p_slots_r[w] = (unsigned)list[4 * w] | ((unsigned)list[4 * w + 1] << 8) |
                                                ((unsigned)list[4 * w + 2] << 16) | ((unsigned)list[4 * w + 3] << 24);
        for (int t = lane; t < ms; t += 64) {
            unsigned word = 0;
            if (t < h.m) {
                const unsigned old = gt_ent[h.k0 + t];
                word = ((unsigned)map[old >> 16] << 16) | (old & 0xffffu);
            }
            p_ent_r[t] = word;
        }
        const int pw = ms * N / 4;
        for (int w = lane; w < pw; w += 64) {
            unsigned word = 0;
            for (int k = 0; k < 4; ++k) {
                const int it = 4 * w + k;
                if (it < h.m * N) word |= (unsigned)gt_pos[(size_t)h.k0 * N + it] << (8 * k);
            }
            p_pos_r[w] = word;
        }
        for (int i = lane; i <= nbs; i += 64)
            p_noff_r[i] = (i <= h.nb) ? (int)(noff[h.i0 + i] - (unsigned)h.r0) : 0;
        __syncthreads();
        for (int s_ = lane; s_ < us; s_ += 64) slot_elem[s_] = new_elem[s_];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ gather tables
// One workgroup per node block.  PASS 0: count the unique adjacent elements of the block (hdr.U);
// PASS 1: write the unique element list and, per (node, element) entry, the packed word
// (unique slot << 16 | local index a << 8 | block-local node).  Entries of a block are the contiguous
// range n2e[n2e_off[i0] .. n2e_off[i1]) -- ascending element per node, so the tables are deterministic.
// WAVES = 0: one workgroup per block (any size that fits the LDS).  WAVES = 4 (round 5): one WAVEFRONT per block, four blocks per workgroup,
// `stride` ints of LDS each -- a block of the usual size has ~56 entries, and 1.46 million workgroups of 256 threads for them were 14 of the
// 60 ms of the 216^3 mesh's set-up; the wavefront's lanes synchronise through the in-order LDS queue (wave_sync) instead of barriers.
template <int PASS, int WAVES = 0>
__global__ void __launch_bounds__(256) k_build_gather_tables(const unsigned* blk_off, const unsigned* noff, const unsigned* n2e_off,
                                                             const unsigned* n2e, int N, GatherHdr* hdr, const unsigned* u_off,
                                                             unsigned* gt_elems, unsigned* gt_ent, const int* conn,
                                                             const unsigned* ncols, unsigned char* gt_pos, int nblk = 0, int stride = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* ent = reinterpret_cast<int*>(smem) + (WAVES ? (int)(threadIdx.x >> 6) * stride : 0);
    const int b = WAVES ? (int)blockIdx.x * WAVES + (int)(threadIdx.x >> 6) : (int)blockIdx.x;
    const int tid = WAVES ? (int)(threadIdx.x & 63) : (int)threadIdx.x, nt = WAVES ? 64 : (int)blockDim.x;
    if (WAVES && b >= nblk) return;
    auto wg_sync = [&]() {
        if (WAVES) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        else __syncthreads();
    };
    const int i0 = (int)blk_off[b], i1 = (int)blk_off[b + 1];
    const int k0 = (int)n2e_off[i0], k1 = (int)n2e_off[i1];
    const int m = k1 - k0;
    int* first = ent + m;
    int* rank = first + m;
    for (int t = tid; t < m; t += nt) ent[t] = (int)(n2e[k0 + t] / (unsigned)N);
    wg_sync();
    for (int t = tid; t < m; t += nt) {
        const int e = ent[t];
        int f = t;
        for (int x = 0; x < t; ++x)
            if (ent[x] == e) { f = x; break; }
        first[t] = f;
    }
    wg_sync();
    for (int t = tid; t < m; t += nt) {
        if (first[t] == t) {
            int r = 0;
            for (int x = 0; x < t; ++x) r += (first[x] == x);
            rank[t] = r;
        }
    }
    wg_sync();
    if (PASS == 0) {
        if (tid == 0) {
            int U = 0;
            for (int x = 0; x < m; ++x) U += (first[x] == x);
            GatherHdr h;
            h.i0 = i0; h.nb = i1 - i0; h.r0 = (int)noff[i0]; h.nrow = (int)(noff[i1] - noff[i0]);
            h.k0 = k0; h.m = m; h.U = U; h.u_off = 0;
            hdr[b] = h;
        }
        return;
    }
    const unsigned uo = u_off[b];
    if (tid == 0) hdr[b].u_off = uo;
    for (int t = tid; t < m; t += nt) {
        if (first[t] == t) gt_elems[uo + rank[t]] = (unsigned)ent[t];
        // owning node: last i with n2e_off[i] <= k0 + t
        int lo = i0, hi = i1;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((int)n2e_off[mid] <= k0 + t) lo = mid; else hi = mid;
        }
        const unsigned a_loc = n2e[k0 + t] % (unsigned)N;
        gt_ent[k0 + t] = ((unsigned)rank[first[t]] << 16) | (a_loc << 8) | (unsigned)(lo - i0);
        if (gt_pos) {  // column slot of every local node of the element inside the row of the owning node
            const unsigned rb = noff[lo], cnt = noff[lo + 1] - rb;
            for (int Jn = 0; Jn < N; ++Jn)
                gt_pos[(size_t)(k0 + t) * N + Jn] =
                    (unsigned char)find_col(ncols + rb, (int)cnt, (unsigned)conn[(size_t)ent[t] * N + Jn]);
        }
    }
}

// link[i] = 1 if nodes i and i+1 share an element (sorted merge of their adjacency lists; entries are e * n + a).
// Runs of linked nodes are what a structured numbering calls grid lines; the block partition aligns to them.
static __global__ void k_linked_to_next(const unsigned* n2e_off, const unsigned* n2e, int N, int num_nodes, unsigned char* link) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_nodes) return;
    unsigned char r = 0;
    if (i + 1 < num_nodes) {
        unsigned a = n2e_off[i], ae = n2e_off[i + 1], b = ae, be = n2e_off[i + 2];
        while (a < ae && b < be) {
            const unsigned ea = n2e[a] / (unsigned)N, eb = n2e[b] / (unsigned)N;
            if (ea == eb) { r = 1; break; }
            if (ea < eb) ++a; else ++b;
        }
    }
    link[i] = r;
}

// ---- locality order of the nodes (row-owner Tet4 kernel on meshes whose numbering has none: see build_partition)
// Morton key of a vertex: 21 bits per coordinate of its position in the bounding box
static __global__ void k_morton_keys(const double* verts, int N, int D, double lo0, double lo1, double lo2, double s0, double s1, double s2,
                              unsigned long long* keys, unsigned* ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double lo[3] = {lo0, lo1, lo2}, sc[3] = {s0, s1, s2};
    unsigned long long key = 0;
    for (int c = 0; c < D; ++c) {
        double t = (verts[(size_t)i * D + c] - lo[c]) * sc[c];
        t = fmin(fmax(t, 0.0), 2097151.0);
        unsigned long long x = (unsigned long long)t;       // spread the 21 bits to every third position
        x = (x | x << 32) & 0x1f00000000ffffull;
        x = (x | x << 16) & 0x1f0000ff0000ffull;
        x = (x | x << 8) & 0x100f00f00f00f00full;
        x = (x | x << 4) & 0x10c30c30c30c30c3ull;
        x = (x | x << 2) & 0x1249249249249249ull;
        key |= x << c;
    }
    keys[i] = key;
    ids[i] = (unsigned)i;
}
static __global__ void k_invert_perm(const unsigned* v2r, int N, int* r2v) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < N) r2v[v2r[v]] = v;
}
// len[v] = length of row v2r[v] of the offsets `off` (len[N] = 0: input of an exclusive scan)
static __global__ void k_perm_row_lengths(const unsigned* off, const unsigned* v2r, int N, unsigned* len) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < N) { const unsigned r = v2r[v]; len[v] = off[r + 1] - off[r]; }
    if (v == N) len[v] = 0;
}
// dst row v = src row v2r[v] (contents unchanged)
static __global__ void k_perm_copy_rows(const unsigned* off_src, const unsigned* src, const unsigned* v2r, const unsigned* off_dst, unsigned* dst,
                                 int N) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const unsigned r = v2r[v], a = off_src[r], n = off_src[r + 1] - a, b = off_dst[v];
    for (unsigned k = 0; k < n; ++k) dst[b + k] = src[a + k];
}
// first node-level CSR entry of the real row of every node in partition order (v2r == null: the natural order)
static __global__ void k_row_starts(const unsigned* noff, const unsigned* v2r, int N, unsigned* row_real) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < N) row_real[v] = noff[v2r ? v2r[v] : (unsigned)v];
}

static __global__ void k_hdr_counts(const GatherHdr* hdr, int nblk, unsigned* counts) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblk) counts[b] = (unsigned)hdr[b].U;
    if (b == nblk) counts[b] = 0;
}

// ============================================================================================ vector
// f_I = sum_q s P g_I  (elliptic.rs:457-531), scattered with fp64 atomics into out[s node + i]
template <int EK, int OP>
__global__ void __launch_bounds__(256) k_assemble_vector(const KArgs a) {
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    constexpr int D = E::D, N = E::N, S = O::S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout L = make_layout<EK, OP, WHAT_VECTOR>(a.nq, a.ub, 0, 0, false);
    double* lds = reinterpret_cast<double*>(smem);
    int* lds_i = reinterpret_cast<int*>(smem + sizeof(double) * (size_t)L.n_doubles);
    const int tid = threadIdx.x, nt = blockDim.x;
    stage_tables<EK>(a, L, lds);
    const long long w0 = a.work_begin + (long long)blockIdx.x * a.epb;
    const int U = (int)min((long long)a.epb, a.work_end - w0);
    for (int i = tid; i < U; i += nt) lds_i[L.o_uniq + i] = a.labels ? (int)a.labels[w0 + i] : (int)(w0 + i);
    __syncthreads();
    stage_elements<EK, S>(a, L, lds, lds_i, U, true, lds_i + L.o_uniq);
    for (int i = tid; i < U * a.nq; i += nt)
        prologue<EK, OP, WHAT_VECTOR>(a, L, lds, lds_i, i / a.nq, i % a.nq, lds_i + L.o_uniq + i / a.nq);
    __syncthreads();
    for (int it = tid; it < U * N; it += nt) {
        const int u = it / N, I = it % N;
        double f[S];
#pragma unroll
        for (int i = 0; i < S; ++i) f[i] = 0.0;
        const double* qp = lds + L.o_QP + (size_t)u * a.nq * L.qpd;
        for (int q = 0; q < a.nq; ++q, qp += L.qpd) {
            const double* g = qp + I * D;
            const double* sp = qp + N * D;
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int k = 0; k < D; ++k) f[i] = fma(sp[i * D + k], g[k], f[i]);
        }
        const int node = lds_i[L.o_cn + u * N + I];
#pragma unroll
        for (int i = 0; i < S; ++i) atomic_add_f64(a.vec_out + (size_t)node * S + i, f[i]);
    }
}

// ============================================================================================ source
// ElementSourceAssembler (src/assembly/local/source.rs:219-278): f_e = sum_q w |det J| f(x_q) phi(xi_q), scattered
// with fp64 atomics into out[s node + c].  One lane per (element, point) computes w |det J| (only the determinant
// of J enters: no singular-Jacobian error on this path), then one lane per (element, node) sums over the points.
// The source is either density_q * g (GravitySource, fenris-solid/src/gravity_source.rs:57-65; density_q is the
// first parameter of the quadrature table) or values sampled by the caller at the physical points.
// Persistent form of k_assemble_vector for the iso-parametric elements with 4 or 8 nodes: a workgroup walks batches of
// 256 / N consecutive elements and prefetches like k_gather_pipelined -- node indices two batches ahead, vertex
// coordinates and u one batch ahead, in registers, parked in LDS behind the contraction -- so that the dependent
// connectivity -> vertex fetches of a batch no longer stand between its phases (they were a third of the kernel).  One
// thread per (element of the batch, local node) throughout: it fetches that node's data and scatters that node's row.
// NT threads: 256, or 128 when the point records of 256 / N elements do not leave room for two workgroups per CU
template <int EK, int OP, int NT>
__global__ void __launch_bounds__(NT) k_assemble_vector_stream(const KArgs a) {
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    constexpr int D = E::D, N = E::N, S = O::S, EPB = NT / N;
    static_assert(E::NG == N && NT % N == 0, "streamed residual kernel: iso-parametric, 4 or 8 nodes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout L = make_layout<EK, OP, WHAT_VECTOR>(a.nq, EPB, 0, 0, false, 0, 1);  // compact point records
    double* lds = reinterpret_cast<double*>(smem);
    int* lds_i = reinterpret_cast<int*>(smem + sizeof(double) * (size_t)L.n_doubles);
    const int tid = threadIdx.x, G = gridDim.x;
    const int u = tid / N, I = tid % N;
    stage_tables<EK>(a, L, lds);
    const long long total = a.work_end - a.work_begin;
    const long long nbatch = (total + EPB - 1) / EPB;
    // element of this thread's slot in batch b, clamped into the work range (branch-free prefetch)
    auto elem_of = [&](long long b) { return a.work_begin + min(b * EPB + u, total - 1); };
    auto load_node = [&](long long b) { return a.conn[(size_t)elem_of(min(b, nbatch - 1)) * N + I]; };
    double V[D], Uv[S];
    auto load_data = [&](int node) {
#pragma unroll
        for (int c = 0; c < D; ++c) V[c] = a.verts[(size_t)node * D + c];
#pragma unroll
        for (int c = 0; c < S; ++c) Uv[c] = a.u ? a.u[(size_t)node * S + c] : 0.0;
    };
    auto park = [&](long long b) {
#pragma unroll
        for (int c = 0; c < D; ++c) lds[L.o_X + u * L.xs + I * D + c] = V[c];
#pragma unroll
        for (int c = 0; c < S; ++c) lds[L.o_U + u * L.us + I * S + c] = Uv[c];
        if (I == 0) lds_i[L.o_uniq + u] = (int)elem_of(min(b, nbatch - 1));
    };
    long long b = blockIdx.x;
    if (b >= nbatch) return;
    int node_cur = load_node(b);
    load_data(node_cur);
    int node_n1 = load_node(b + G);
    park(b);
    asm volatile("" : "+v"(node_n1));
    __syncthreads();
    for (; b < nbatch; b += G) {
        load_data(node_n1);                    // lands while this batch is computed
        int node_n2 = load_node(b + 2LL * G);
        // phase B: one lane per (element, point)
        for (int i = tid; i < EPB * a.nq; i += NT)
            prologue<EK, OP, WHAT_VECTOR, false, false, true>(a, L, lds, lds_i, i / a.nq, i % a.nq, lds_i + L.o_uniq + i / a.nq);
        lds_barrier();
        // contraction: this thread's node row of the element vector, scattered with fp64 atomics
        {
            double f[S];
#pragma unroll
            for (int i = 0; i < S; ++i) f[i] = 0.0;
            const double* qp = lds + L.o_QP + (size_t)u * a.nq * L.qpd;
            const double* r = lds + L.o_gref + I * D;  // reference gradient of this node, one table row per point
            for (int q = 0; q < a.nq; ++q, qp += L.qpd, r += L.gs) {
#pragma unroll
                for (int i = 0; i < S; ++i)
#pragma unroll
                    for (int m = 0; m < D; ++m) f[i] = fma(qp[i * D + m], r[m], f[i]);
            }
            // The prefetched batch is parked BEFORE this batch's stores: loads and stores share one in-order counter, so a
            // wait for the loads behind the stores would wait for the stores as well (a write latency per batch).  X / U
            // are read by phase B only, which lies behind the barrier above.
            park(b + G);
            if (b * EPB + u < total) {
                if (a.ke_out) {  // two-pass form: the element vectors, (element, local node, component); k_vector_from_elements sums
                    double* dst = a.ke_out + ((size_t)elem_of(b) * N + I) * S;
#pragma unroll
                    for (int i = 0; i < S; ++i) dst[i] = f[i];
                } else {
#pragma unroll
                    for (int i = 0; i < S; ++i) atomic_add_f64(a.vec_out + (size_t)node_cur * S + i, f[i]);
                }
            }
        }
        asm volatile("" : "+v"(node_n2));
        node_cur = node_n1;
        node_n1 = node_n2;
        lds_barrier();
    }
}

// second pass of the atomics-free residual: out[s node + c] += sum over the node's (element, local node) entries, in ascending
// entry order (n2e is sorted: element-major like the reference's sequential loop) -- bitwise reproducible
template <int S>
__global__ void __launch_bounds__(256) k_vector_from_elements(int num_nodes, const unsigned* n2e_off, const unsigned* n2e, const double* fe,
                                                              double* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)num_nodes * S) return;
    const int node = (int)(i / S), c = (int)(i % S);
    double acc = 0.0;
    for (unsigned k = n2e_off[node]; k < n2e_off[node + 1]; ++k) acc += fe[(size_t)n2e[k] * S + c];
    out[i] += acc;
}

struct SourceArgs {
    int N, NG;                 // nodes per element (solution / geometry; the geometry nodes come first)
    const double* phigeom;     // nq x NG  basis values of the geometry map
    const double* g;           // S (device) or null
    const double* values;      // E x nq x S (device) or null
    double* xq;                // E x nq x D physical points (k_physical_points)
};

template <int D>
__device__ __forceinline__ double source_wdet(const KArgs& a, const SourceArgs& sa, long long e, int q, double* x_out) {
    double J[D][D], x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        x[i] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) J[i][j] = 0.0;
    }
    for (int g = 0; g < sa.NG; ++g) {
        const double* v = a.verts + (size_t)a.conn[(size_t)e * sa.N + g] * D;
        const double* gg = a.ggeom + ((size_t)q * sa.NG + g) * D;
        const double ph = sa.phigeom[(size_t)q * sa.NG + g];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            x[i] = fma(v[i], ph, x[i]);
#pragma unroll
            for (int j = 0; j < D; ++j) J[i][j] = fma(v[i], gg[j], J[i][j]);
        }
    }
    if (x_out)
#pragma unroll
        for (int i = 0; i < D; ++i) x_out[i] = x[i];
    return a.qw[q] * fabs(det_small<D>(J));
}

template <int D, int S>
__global__ void __launch_bounds__(256) k_assemble_source(const KArgs a, const SourceArgs sa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* wd = reinterpret_cast<double*>(smem);  // epb x nq
    const int tid = threadIdx.x, nt = blockDim.x;
    const long long w0 = a.work_begin + (long long)blockIdx.x * a.epb;
    const int U = (int)min((long long)a.epb, a.work_end - w0);
    for (int i = tid; i < U * a.nq; i += nt) {
        const long long e = a.labels ? (long long)a.labels[w0 + i / a.nq] : w0 + i / a.nq;
        wd[i] = source_wdet<D>(a, sa, e, i % a.nq, nullptr);
    }
    __syncthreads();
    for (int it = tid; it < U * sa.N; it += nt) {
        const int u = it / sa.N, I = it % sa.N;
        const long long e = a.labels ? (long long)a.labels[w0 + u] : w0 + u;
        double f[S];
#pragma unroll
        for (int c = 0; c < S; ++c) f[c] = 0.0;
        for (int q = 0; q < a.nq; ++q) {
            const double t = wd[u * a.nq + q] * a.phiref[(size_t)q * sa.N + I];
#pragma unroll
            for (int c = 0; c < S; ++c) {
                const double rho = a.rule_map ? a.rparams[((size_t)a.rule_map[e] * a.nq + q) * 2] : (a.qparams ? a.qparams[2 * q] : 0.0);
                const double fc = sa.values ? sa.values[((size_t)e * a.nq + q) * S + c] : sa.g[c] * rho;
                f[c] = fma(t, fc, f[c]);
            }
        }
        if (a.ke_out) {  // two-pass form (see k_vector_from_elements): element vectors to scratch
            double* dst = a.ke_out + ((size_t)e * sa.N + I) * S;
#pragma unroll
            for (int c = 0; c < S; ++c) dst[c] = f[c];
        } else {
            const int node = a.conn[(size_t)e * sa.N + I];
#pragma unroll
            for (int c = 0; c < S; ++c) atomic_add_f64(a.vec_out + (size_t)node * S + c, f[c]);
        }
    }
}

// x_q = map_reference_coords(xi_q) of every element (source.rs:263), E x nq x D
template <int D>
__global__ void __launch_bounds__(256) k_physical_points(const KArgs a, const SourceArgs sa) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.num_elements * a.nq) return;
    double x[D];
    (void)source_wdet<D>(a, sa, i / a.nq, (int)(i % a.nq), x);
#pragma unroll
    for (int k = 0; k < D; ++k) sa.xq[(size_t)i * D + k] = x[k];
}

// ============================================================================================ scalar
template <int EK, int OP>
__global__ void __launch_bounds__(256) k_assemble_scalar(const KArgs a) {
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    constexpr int S = O::S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout L = make_layout<EK, OP, WHAT_SCALAR>(a.nq, a.ub, 0, 0, false);
    double* lds = reinterpret_cast<double*>(smem);
    int* lds_i = reinterpret_cast<int*>(smem + sizeof(double) * (size_t)L.n_doubles);
    const int tid = threadIdx.x, nt = blockDim.x;
    stage_tables<EK>(a, L, lds);
    const long long w0 = a.work_begin + (long long)blockIdx.x * a.epb;
    const int U = (int)min((long long)a.epb, a.work_end - w0);
    for (int i = tid; i < U; i += nt) lds_i[L.o_uniq + i] = a.labels ? (int)a.labels[w0 + i] : (int)(w0 + i);
    __syncthreads();
    stage_elements<EK, S>(a, L, lds, lds_i, U, true, lds_i + L.o_uniq);
    for (int i = tid; i < U * a.nq; i += nt)
        prologue<EK, OP, WHAT_SCALAR>(a, L, lds, lds_i, i / a.nq, i % a.nq, lds_i + L.o_uniq + i / a.nq);
    __syncthreads();
    // element energies (sum over the points, like compute_element_elliptic_energy), then one partial per block summed in
    // element order; the host adds the partials in block order => deterministic
    if (tid < U) {
        double t = 0.0;
        for (int q = 0; q < a.nq; ++q) t += lds[L.o_QP + tid * a.nq + q];
        lds[L.o_QP + tid * a.nq] = t;
    }
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int i = 0; i < U; ++i) tot += lds[L.o_QP + i * a.nq];
        a.scalar_out[blockIdx.x] = tot;
    }
}

}  // namespace fenris_hip
