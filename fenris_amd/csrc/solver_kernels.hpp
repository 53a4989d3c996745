// Callers of the assembly path that keep K on the device: Jacobi-preconditioned conjugate
// gradients on the node-blocked CSR (fenris-sparse/src/cg.rs:196-474 as driven by solve_linear_system,
// tests/convergence_tests/poisson_mms_common.rs:142-163) and the L2 / H1 error integrals (src/error.rs:287-372).
// All reductions are two-stage: per-workgroup partial sums in a fixed order, summed in workgroup order by the
// host => bitwise reproducible runs (no floating-point atomics).
#pragma once
#include <hip/hip_runtime.h>

#include "assemble_kernels.hpp"
#include "device_common.hpp"

namespace fenris_hip {

// sum over the 64 lanes of a wavefront (butterfly, every lane gets the total)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum of K per-thread values over a 256-thread workgroup, in a fixed order; thread 0 stores to out[0..K).
template <int K>
__device__ __forceinline__ void block_sum_store(double (&v)[K], double* out) {
    __shared__ double part[4][K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) part[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = (part[0][k] + part[1][k]) + (part[2][k] + part[3][k]);
}

// `count` groups of K partials -> gridDim.x groups of K sums, each over one contiguous part of them, in a fixed order: the SpMV and the vector kernels of
// the CG loop launch short-lived workgroups (see spmv_launch) and leave K partials per workgroup; the host sums what this kernel leaves
template <int K>
static __global__ void __launch_bounds__(256) k_sum_partial_ranges(const double* in, long long count, double* out) {
    const long long per = (count + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = min(count, lo + per);
    double s[K];
#pragma unroll
    for (int j = 0; j < K; ++j) s[j] = 0.0;
    for (long long k = lo + threadIdx.x; k < hi; k += 256)
#pragma unroll
        for (int j = 0; j < K; ++j) s[j] += in[K * k + j];
    block_sum_store<K>(s, out + K * blockIdx.x);
}

// y = A x on the node-blocked CSR: the S rows of node i are contiguous, each S * cnt long, and share the
// node-level column list (global.rs:100-118).  One wavefront per node (grid-stride); lanes stride over the
// row entries, S row sums per lane, butterfly reduction.  Optionally the partial of  x . y  per workgroup.
template <int S>
__global__ void __launch_bounds__(256) k_spmv_blocked(int num_nodes, const unsigned* noff, const unsigned* ncols, const double* vals,
                                                      const double* x, double* y, double* dot_partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double dot[1] = {0.0};
    for (int i = blockIdx.x * 4 + wave; i < num_nodes; i += gridDim.x * 4) {
        const unsigned r0 = noff[i], cnt = noff[i + 1] - r0;
        const int len = S * (int)cnt;
        const double* row = vals + (size_t)S * S * r0;
        double acc[S];
#pragma unroll
        for (int a = 0; a < S; ++a) acc[a] = 0.0;
        for (int t = lane; t < len; t += 64) {
            const double xv = x[(size_t)S * ncols[r0 + t / S] + t % S];
#pragma unroll
            for (int a = 0; a < S; ++a) acc[a] = fma(row[(size_t)a * len + t], xv, acc[a]);
        }
#pragma unroll
        for (int a = 0; a < S; ++a) {
            const double s = wave_sum(acc[a]);
            if (lane == 0) {
                y[(size_t)S * i + a] = s;
                dot[0] = fma(x[(size_t)S * i + a], s, dot[0]);
            }
        }
    }
    if (dot_partial) block_sum_store<1>(dot, dot_partial + blockIdx.x);
}

// sum over the 32 lanes of a half wavefront by DPP moves on the vector pipe (xor 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast:15 hands the
// total of rows 0 / 2 to rows 1 / 3): the lanes 16 .. 31 of the half hold the total.  Through __shfl_xor the same sum is ten ds_bpermute per value --
// LDS-pipe instructions; the SpMV issued sixty of them per four nodes.
__device__ __forceinline__ double half_sum_hi(double v) {
    v += dpp_quad_full<0xB1>(v);
    v += dpp_quad_full<0x4E>(v);
    v += dpp_xor4(v);
    {   // row_ror:8 = lane ^ 8 inside a row of 16
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = __builtin_amdgcn_mov_dpp((int)b, 0x128, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x128, 0xF, 0xF, true);
        v += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
    }
    {   // row_bcast:15 into rows 1 and 3 (row mask 0xA); rows 0 and 2 add zero
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x142, 0xA, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x142, 0xA, 0xF, false);
        v += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
    }
    return v;
}

// The same product for patterns whose node rows hold at most 32 column blocks (Quad4 9, Tet4 ~15, Hex8 27): HALF a wavefront per
// node, one lane per column block -- the lane fetches its column index, the S entries of x and its S x S block (S runs of S
// contiguous doubles; neighbouring lanes read neighbouring runs) -- and two node pairs per wavefront in flight.  The
// one-wavefront-per-node form above walks a row of 81 entries in two trips of 64 lanes behind three dependent fetches (row offset
// -> column index -> x): with 32 wavefronts per CU that chain, not the memory system, set its rate (3.2 TB/s on Hex8 216^3).
template <int S, int UN = 2>
__global__ void __launch_bounds__(256) k_spmv_blocked_half(int num_nodes, const unsigned* noff, const unsigned* ncols, const double* vals,
                                                           const double* x, double* y, double* dot_partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
    // UN: node pairs in flight per wavefront
    double dot[1] = {0.0};
    const int stride = gridDim.x * 4 * 2 * UN;
    for (int base = (blockIdx.x * 4 + wave) * 2 * UN; base < num_nodes; base += stride) {
        double acc[UN][S];
        int node[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            node[u] = base + 2 * u + half;
            const int nc = min(node[u], num_nodes - 1);
            const unsigned r0 = noff[nc], cnt = noff[nc + 1] - r0;
            const bool act = node[u] < num_nodes && (unsigned)hl < cnt;
#pragma unroll
            for (int a = 0; a < S; ++a) acc[u][a] = 0.0;
            if (act) {
                const unsigned col = ncols[r0 + hl];
                double xv[S];
#pragma unroll
                for (int c = 0; c < S; ++c) xv[c] = x[(size_t)S * col + c];
                const double* blk = vals + (size_t)S * S * r0 + (size_t)S * hl;
                if constexpr (S == 3) {
                    // a row of the block is 24 bytes at an 8-byte boundary: one 16-byte and one 8-byte load (six load instructions per lane instead of nine)
                    typedef double f64x2_u8s __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
                    for (int a = 0; a < S; ++a) {
                        const double* rw = blk + (size_t)a * S * cnt;
                        const f64x2_u8s v01 = *reinterpret_cast<const f64x2_u8s*>(rw);
                        const double v2 = rw[2];
                        acc[u][a] = fma(v01.x, xv[0], acc[u][a]);
                        acc[u][a] = fma(v01.y, xv[1], acc[u][a]);
                        acc[u][a] = fma(v2, xv[2], acc[u][a]);
                    }
                } else {
#pragma unroll
                for (int a = 0; a < S; ++a)
#pragma unroll
                    for (int c = 0; c < S; ++c) acc[u][a] = fma(blk[(size_t)a * S * cnt + c], xv[c], acc[u][a]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int a = 0; a < S; ++a) {
                const double s = half_sum_hi(acc[u][a]);   // over the half wavefront: its lanes 16 .. 31 hold the sum
                if (hl == 16 && node[u] < num_nodes) {
                    y[(size_t)S * node[u] + a] = s;
                    dot[0] = fma(x[(size_t)S * node[u] + a], s, dot[0]);
                }
            }
    }
    if (dot_partial) block_sum_store<1>(dot, dot_partial + blockIdx.x);
}

// 1 / diagonal of the blocked CSR (matrix.diagonal_as_csr() + recip, poisson_mms_common.rs:148-151)
template <int S>
__global__ void __launch_bounds__(256) k_inverse_diagonal(int num_nodes, const unsigned* noff, const unsigned* ncols, const double* vals,
                                                          double* dinv) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= S * num_nodes) return;
    const int i = r / S, a = r % S;
    const unsigned r0 = noff[i], cnt = noff[i + 1] - r0;
    const int pos = find_col(ncols + r0, (int)cnt, (unsigned)i);
    dinv[r] = 1.0 / vals[(size_t)S * S * r0 + (size_t)a * S * cnt + (size_t)S * pos + a];
}

// r = b - r (r holds A x),  z = M^-1 r,  p = z;  partials of (z.r, b.b, r.r)      cg.rs:388-404
static __global__ void __launch_bounds__(256) k_cg_init(int n, const double* b, const double* dinv, double* r, double* z, double* p,
                                                 double* partial /* gridDim.x x 3 */) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double ri = b[i] - r[i];
        const double zi = dinv ? dinv[i] * ri : ri;
        r[i] = ri;
        z[i] = zi;
        p[i] = zi;
        s[0] = fma(zi, ri, s[0]);
        s[1] = fma(b[i], b[i], s[1]);
        s[2] = fma(ri, ri, s[2]);
    }
    block_sum_store<3>(s, partial + 3 * blockIdx.x);
}

// x += alpha p,  r -= alpha Ap,  z = M^-1 r;  partials of (z.r, r.r)               cg.rs:453-468
static __global__ void __launch_bounds__(256) k_cg_update(int n, double alpha, const double* p, const double* Ap, const double* dinv,
                                                   double* x, double* r, double* z, double* partial /* gridDim.x x 2 */) {
    double s[2] = {0.0, 0.0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i];
        const double zi = dinv ? dinv[i] * ri : ri;
        r[i] = ri;
        z[i] = zi;
        s[0] = fma(zi, ri, s[0]);
        s[1] = fma(ri, ri, s[1]);
    }
    block_sum_store<2>(s, partial + 2 * blockIdx.x);
}

// p = beta p + z                                                                   cg.rs:470-474
static __global__ void __launch_bounds__(256) k_cg_direction(int n, double beta, const double* z, double* p) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = beta * p[i] + z[i];
}

// ---- error integrals (src/error.rs:287-372): one lane per (element, quadrature point)
//   WHICH 0:  w |det J| |u_h(x_q) - u(x_q)|^2                      make_L2_error_squared_integrand   :222-236
//   WHICH 1:  w |det J| |grad u_h(x_q) - grad u(x_q)|_F^2          make_H1_seminorm_error_squared_integrand :238-254
// exact[] holds the reference solution sampled by the caller at the physical points:  (E, nq, S) values, or
// (E, nq, D, S) gradients with grad[i][k] = d u_k / d x_i (OMatrix<GeometryDim, SolutionDim>).
template <int D, int S, int WHICH>
__global__ void __launch_bounds__(256) k_error_squared(const KArgs a, const SourceArgs sa, const double* uh, const double* exact,
                                                       double* partial) {
    double s[1] = {0.0};
    const long long total = a.num_elements * a.nq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i / a.nq;
        const int q = (int)(i % a.nq);
        const int* nodes = a.conn + (size_t)e * sa.N;
        double J[D][D];
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int c = 0; c < D; ++c) J[r][c] = 0.0;
        for (int g = 0; g < sa.NG; ++g) {
            const double* v = a.verts + (size_t)nodes[g] * D;
            const double* gg = a.ggeom + ((size_t)q * sa.NG + g) * D;
#pragma unroll
            for (int r = 0; r < D; ++r)
#pragma unroll
                for (int c = 0; c < D; ++c) J[r][c] = fma(v[r], gg[c], J[r][c]);
        }
        const double detJ = det_small<D>(J);
        double err2 = 0.0;
        if (WHICH == 0) {
            double u[S];
#pragma unroll
            for (int k = 0; k < S; ++k) u[k] = 0.0;
            for (int n = 0; n < sa.N; ++n) {
                const double ph = a.phiref[(size_t)q * sa.N + n];
#pragma unroll
                for (int k = 0; k < S; ++k) u[k] = fma(ph, uh[(size_t)nodes[n] * S + k], u[k]);
            }
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const double d = u[k] - exact[(size_t)i * S + k];
                err2 = fma(d, d, err2);
            }
        } else {
            double Ji[D][D];
            if (detJ == 0.0) {
                report_singular(a.status, e);
                continue;
            }
            inv_small(J, detJ, Ji);
            double gu[D][S];
#pragma unroll
            for (int r = 0; r < D; ++r)
#pragma unroll
                for (int k = 0; k < S; ++k) gu[r][k] = 0.0;
            for (int n = 0; n < sa.N; ++n) {
                const double* gr = a.gref + ((size_t)q * sa.N + n) * D;
                double g[D];
#pragma unroll
                for (int r = 0; r < D; ++r) {
                    double t = 0.0;
#pragma unroll
                    for (int c = 0; c < D; ++c) t = fma(Ji[c][r], gr[c], t);  // J^-T grad_ref
                    g[r] = t;
                }
#pragma unroll
                for (int r = 0; r < D; ++r)
#pragma unroll
                    for (int k = 0; k < S; ++k) gu[r][k] = fma(g[r], uh[(size_t)nodes[n] * S + k], gu[r][k]);
            }
#pragma unroll
            for (int r = 0; r < D; ++r)
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const double d = gu[r][k] - exact[((size_t)i * D + r) * S + k];
                    err2 = fma(d, d, err2);
                }
        }
        s[0] = fma(a.qw[q] * fabs(detJ), err2, s[0]);
    }
    block_sum_store<1>(s, partial + blockIdx.x);
}

}  // namespace fenris_hip
