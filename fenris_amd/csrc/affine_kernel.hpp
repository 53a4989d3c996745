// Affine-element classification for the owner-computes stiffness kernel k_affine_rows (affine_rows.hip; Hex8, Laplace /
// uniform LinearElastic): the per-element test and the per-block class.  The kernel itself and its tables live in
// affine_rows.hip / affine_rows.hpp.
//
// Which elements qualify is decided per element from the vertex coordinates (k_classify_affine_hex8); node blocks all of
// whose elements qualify run on k_affine_rows, the others keep the general kernels (engine.hip: build_partition splits the
// sweep order by class).
#pragma once
#include <hip/hip_runtime.h>

#include "assemble_kernels.hpp"
#include "device_common.hpp"
#include "rows_kernel.hpp"

namespace fenris_hip {

constexpr int AFFINE_GW_LE = 10, AFFINE_GW_LAP = 6;  // doubles per reference block in fh_ctx::ghat (LinearElastic | Laplace)

// 1 if the trilinear map of a hexahedron is affine to the relative tolerance `tol`: with the node signs of
// hexahedron.rs:49-58 the map is  c0 + c1 xi + c2 eta + c3 zeta + c12 xi eta + c23 eta zeta + c31 zeta xi + c123 xi eta zeta,
// c_* = 1/8 sum_a sign X_a; affine iff the four mixed coefficients vanish.  Compared against the shortest of c1, c2, c3.
static __global__ void __launch_bounds__(256) k_classify_affine_hex8(const double* verts, const int* conn, long long E, double tol,
                                                              unsigned char* out, unsigned long long* count) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = e < E;
    bool ok = false;
    if (in_range) {
    const int sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    double c[7][3];
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int i = 0; i < 3; ++i) c[k][i] = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const double* X = verts + (size_t)conn[e * 8 + g] * 3;
        const int s[7] = {sx[g], sy[g], sz[g], sx[g] * sy[g], sy[g] * sz[g], sz[g] * sx[g], sx[g] * sy[g] * sz[g]};
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int i = 0; i < 3; ++i) c[k][i] += s[k] * X[i];
    }
    double lin_min = 1e300, dev = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) lin_min = fmin(lin_min, sqrt(c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2]));
#pragma unroll
    for (int k = 3; k < 7; ++k) dev = fmax(dev, sqrt(c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2]));
    ok = lin_min > 0.0 && dev <= tol * lin_min;  // NaN / inf coordinates compare false: general path
    out[e] = ok ? 1 : 0;
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}

static __global__ void __launch_bounds__(256) k_bytes_differ(const unsigned char* x, const unsigned char* y, long long n, int* differ) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && x[i] != y[i]) *differ = 1;
}

// class of a node block: 1 if every adjacent element is affine (and the block fits the affine kernel's slot budget)
static __global__ void __launch_bounds__(256) k_block_class(const GatherHdr* hdr, const unsigned* gt_elems, const unsigned char* elem_aff,
                                                     int nblk, int max_u, unsigned char* cls) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const GatherHdr h = hdr[b];
    unsigned char ok = (h.U <= max_u) ? 1 : 0;
    for (int k = 0; k < h.U && ok; ++k) ok = elem_aff[gt_elems[h.u_off + k]];
    cls[b] = ok;
}

// smallest and largest non-negative entry (mm = {INT_MAX, -1} on entry): the element range behind a set of position tables
static __global__ void __launch_bounds__(256) k_minmax_nonneg(const int* v, size_t n, int* mm) {
    int lo = 0x7fffffff, hi = -1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = v[i];
        if (x >= 0) { lo = min(lo, x); hi = max(hi, x); }
    }
    if (hi >= 0) { atomicMin(mm, lo); atomicMax(mm + 1, hi); }
}

}  // namespace fenris_hip
