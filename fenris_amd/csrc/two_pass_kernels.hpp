// Second pass of the two-pass owner-computes assembly (engine_two_pass.hip): the row gather from dense element matrices, and the
// per-pattern tables it walks.  Split from assemble_kernels.hpp in round 5 (only engine_two_pass.hip includes it).
#pragma once
#include <hip/hip_runtime.h>

#include "assemble_kernels.hpp"
#include "device_common.hpp"

namespace fenris_hip {

// ============================================================================================ rows from dense K_e
// Second pass of the two-pass owner-computes assembly used for high-order elements (n > 8): the dense element
// matrices were written by k_assemble_matrix<MODE_DUMP> (column-major, both triangles); here one wavefront owns a
// node, walks the node's (element, local index a) entries and adds the columns S a .. S a + S - 1 of K_e -- by
// symmetry its rows, but contiguous -- into the node's CSR rows held in LDS, then writes the rows once, coalesced.
// Every K_e entry is read exactly once and every CSR value written exactly once; no atomics (the lanes of a
// wavefront hit distinct targets within an entry, and a wavefront's LDS operations execute in order).
// column slot of every local node of every (node, element) entry inside the owning node's row (one byte or one
// 16-bit word per (entry, local node)); built once per pattern for the two-pass assembly
template <typename PT>
__global__ void __launch_bounds__(256) k_entry_positions(long long total, int n, const unsigned* adj_off, const unsigned* adj,
                                                         const unsigned* noff, const unsigned* ncols, const int* conn,
                                                         const int* entry_node, PT* pos) {
    const long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= total) return;
    const long long t = it / n;
    const int J = (int)(it % n);
    const int i = entry_node[t];
    const unsigned ent = adj[t];
    const unsigned r0 = noff[i];
    pos[it] = (PT)find_col(ncols + r0, (int)(noff[i + 1] - r0), (unsigned)conn[(size_t)(ent / (unsigned)n) * n + J]);
    (void)adj_off;
}
static __global__ void __launch_bounds__(256) k_entry_nodes(int num_nodes, const unsigned* adj_off, int* entry_node) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_nodes) return;
    for (unsigned t = adj_off[i]; t < adj_off[i + 1]; ++t) entry_node[t] = i;
}

// k_rows_from_dense for small column-major element matrices (S n <= P <= 32, P a power of two): a row of K_e fills less
// than half a wavefront, so 64 / P entries of the node share one load instruction (lane / P picks the entry) and all
// EB groups of a node are in flight together -- Hex8: the 8 entries of a node in one round (the one-entry-per-load
// form ran this pass at 2.7 TB/s with 24 of 64 lanes busy).
template <int S, typename PT, int P>
__global__ void __launch_bounds__(256) k_rows_from_dense_small(int num_nodes, int n, const unsigned* noff, const unsigned* adj_off,
                                                               const unsigned* adj, const PT* pos_tab, const double* ke, double* vals,
                                                               int overwrite, int max_cnt, const int* node_list, int node_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = (int)(blockDim.x >> 6);
    double* acc = reinterpret_cast<double*>(smem) + (size_t)wave * S * S * max_cnt;
    const int ld = S * n;
    constexpr int G = 64 / P, EB = 4;
    const int sub = lane / P, idx = min(lane % P, ld - 1);
    const bool lane_in_row = (lane % P) < ld;
    (void)num_nodes;
    for (int it = blockIdx.x * wpb + wave; it < node_count; it += gridDim.x * wpb) {
        const int i = node_list ? node_list[it] : it;
        const unsigned r0 = noff[i];
        const int cnt = (int)(noff[i + 1] - r0);
        for (int k = lane; k < S * S * cnt; k += 64) acc[k] = 0.0;
        const unsigned t0 = adj_off[i], t1 = adj_off[i + 1];
        for (unsigned t = t0; t < t1; t += EB * G) {
            double v[EB][S];
            int pos[EB];
            bool ok[EB];
#pragma unroll
            for (int k = 0; k < EB; ++k) {
                const unsigned tk = t + (unsigned)(k * G + sub);
                ok[k] = tk < t1 && lane_in_row;
                const unsigned tc = min(tk, t1 - 1);
                const unsigned ent = adj[tc];
                const int e = (int)(ent / (unsigned)n), a = (int)(ent % (unsigned)n);
                const double* kb = ke + (size_t)e * ld * ld + (size_t)S * a * ld;  // column S a + r of the symmetric K_e
                pos[k] = (int)pos_tab[(size_t)tc * n + idx / S];
#pragma unroll
                for (int r = 0; r < S; ++r) v[k][r] = kb[(size_t)r * ld + idx];
            }
#pragma unroll
            for (int k = 0; k < EB; ++k)
                if (ok[k]) {
                    double* dst = acc + S * pos[k] + idx % S;
#pragma unroll
                    for (int r = 0; r < S; ++r) atomic_add_f64(dst + r * S * cnt, v[k][r]);
                }
        }
        double* out = vals + (size_t)S * S * r0;
        if (overwrite) for (int k = lane; k < S * S * cnt; k += 64) out[k] = acc[k];
        else for (int k = lane; k < S * S * cnt; k += 64) out[k] += acc[k];
    }
}

template <int S, typename PT>
__global__ void __launch_bounds__(256) k_rows_from_dense(int num_nodes, int n, const unsigned* noff, const unsigned* adj_off,
                                                         const unsigned* adj, const PT* pos_tab, const double* ke, double* vals,
                                                         int overwrite, int max_cnt, const int* node_list, int node_count) {
    // node_list: a subset / another order of the nodes instead of all nodes in order (or null); the workgroup may be a single wavefront
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = (int)(blockDim.x >> 6);
    double* acc = reinterpret_cast<double*>(smem) + (size_t)wave * S * S * max_cnt;
    const int ld = S * n;
    (void)num_nodes;
    for (int it = blockIdx.x * wpb + wave; it < node_count; it += gridDim.x * wpb) {
        const int i = __builtin_amdgcn_readfirstlane(node_list ? node_list[it] : it);
        const unsigned r0 = noff[i];
        const int cnt = (int)(noff[i + 1] - r0);
        for (int k = lane; k < S * S * cnt; k += 64) acc[k] = 0.0;
        // Groups of EB entries: (element, local index) are wave-uniform (scalar loads); all K_e loads of the group
        // (clamped, branch-free) are issued before the first LDS add, so that EB x 2 x S loads per lane are in
        // flight -- with the loads of one entry at a time the pass ran at a quarter of the HBM rate.
        // Targets of different entries may coincide (two elements sharing a neighbour node) => LDS atomics.
        constexpr int EB = 4, HB = 2;  // entries per group, 64-lane column halves (S n <= 128 per half pair)
        const unsigned t0 = __builtin_amdgcn_readfirstlane(adj_off[i]), t1 = __builtin_amdgcn_readfirstlane(adj_off[i + 1]);
        for (unsigned t = t0; t < t1; t += EB) {
            for (int h0 = 0; h0 * 64 < ld; h0 += HB) {
                double v[EB][HB][S];
                int pos[EB][HB];
#pragma unroll
                for (int k = 0; k < EB; ++k) {
                    const unsigned tk = min(t + (unsigned)k, t1 - 1);
                    const unsigned ent = __builtin_amdgcn_readfirstlane(adj[tk]);
                    const int e = (int)(ent / (unsigned)n), a = (int)(ent % (unsigned)n);
                    const double* kb = ke + (size_t)e * ld * ld + (size_t)S * a * ld;
                    const PT* pp = pos_tab + (size_t)tk * n;
#pragma unroll
                    for (int h = 0; h < HB; ++h) {
                        const int idx = min(lane + 64 * (h0 + h), ld - 1);
                        pos[k][h] = (int)pp[idx / S];
#pragma unroll
                        for (int r = 0; r < S; ++r)
                            v[k][h][r] = kb[(size_t)r * ld + idx];
                    }
                }
#pragma unroll
                for (int k = 0; k < EB; ++k)
#pragma unroll
                    for (int h = 0; h < HB; ++h) {
                        const int idx = lane + 64 * (h0 + h);
                        if (t + (unsigned)k < t1 && idx < ld) {
                            double* dst = acc + S * pos[k][h] + idx % S;
#pragma unroll
                            for (int r = 0; r < S; ++r) atomic_add_f64(dst + r * S * cnt, v[k][h][r]);
                        }
                    }
            }
        }
        double* out = vals + (size_t)S * S * r0;
        if (overwrite) for (int k = lane; k < S * S * cnt; k += 64) out[k] = acc[k];
        else for (int k = lane; k < S * S * cnt; k += 64) out[k] += acc[k];
    }
}


// Second pass for the Hex27 matrix-core first pass (hex27_blocks.hpp): the element matrices are stored as their UPPER NODE-BLOCK TRIANGLE,
// ke[e][tri(I, J)][i][j] with tri(I, J) = I (53 - I) / 2 + J for I <= J (378 blocks of 3 x 3 = 3 402 doubles instead of 6 561).  The rows of local
// node a of element e are the blocks (a, J), J >= a -- one contiguous run -- and, by symmetry (K_e[(a, r), (J, c)] = K_e[(J, c), (a, r)]), the
// TRANSPOSED blocks (J, a), J < a: a 72-byte piece each.  A lane takes the entries idx = lane + 64 h, h < 4, of the entry's 243 values in the
// order (J, r, c).  Both halves of a symmetric pair of the global matrix are sums of the SAME stored doubles in the same (ascending element)
// order: the assembled matrix is symmetric bit for bit.
template <typename PT>
__global__ void __launch_bounds__(256) k_rows_from_tri(const unsigned* noff, const unsigned* adj_off, const unsigned* adj, const PT* pos_tab,
                                                       const double* ke, double* vals, int overwrite, int max_cnt, const int* node_list,
                                                       int node_count) {
    constexpr int S = 3, n = 27, TRI = (n * (n + 1) / 2) * 9, EB = 4, HR = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = (int)(blockDim.x >> 6);
    double* acc = reinterpret_cast<double*>(smem) + (size_t)wave * S * S * max_cnt;
    // the lane's four places inside an entry (the last round is partly empty: 243 = 3 x 64 + 51)
    int Jl[HR], rcl[HR], tl[HR];
#pragma unroll
    for (int h = 0; h < HR; ++h) {
        const int idx = min(lane + 64 * h, S * S * n - 1);
        Jl[h] = idx / 9;
        rcl[h] = idx - 9 * Jl[h];
        tl[h] = (rcl[h] % 3) * 3 + rcl[h] / 3;   // the same place in the transposed block
    }
    for (int it = blockIdx.x * wpb + wave; it < node_count; it += gridDim.x * wpb) {
        const int i = __builtin_amdgcn_readfirstlane(node_list ? node_list[it] : it);
        const unsigned r0 = noff[i];
        const int cnt = (int)(noff[i + 1] - r0);
        for (int k = lane; k < S * S * cnt; k += 64) acc[k] = 0.0;
        const unsigned t0 = __builtin_amdgcn_readfirstlane(adj_off[i]), t1 = __builtin_amdgcn_readfirstlane(adj_off[i + 1]);
        for (unsigned t = t0; t < t1; t += EB) {
            double v[EB][HR];
            int pos[EB][HR];
#pragma unroll
            for (int k = 0; k < EB; ++k) {
                const unsigned tk = min(t + (unsigned)k, t1 - 1);
                const unsigned ent = __builtin_amdgcn_readfirstlane(adj[tk]);
                const int e = (int)(ent / (unsigned)n), a = (int)(ent % (unsigned)n);
                const double* kb = ke + (size_t)e * TRI;
                const PT* pp = pos_tab + (size_t)tk * n;
                const int row_a = (a * (53 - a)) / 2;
#pragma unroll
                for (int h = 0; h < HR; ++h) {
                    const int J = Jl[h];
                    const int src = J >= a ? (row_a + J) * 9 + rcl[h] : ((J * (53 - J)) / 2 + a) * 9 + tl[h];
                    pos[k][h] = (int)pp[J];
                    v[k][h] = kb[src];
                }
            }
#pragma unroll
            for (int k = 0; k < EB; ++k)
#pragma unroll
                for (int h = 0; h < HR; ++h)
                    if (t + (unsigned)k < t1 && lane + 64 * h < S * S * n)
                        atomic_add_f64(acc + (rcl[h] / 3) * S * cnt + S * pos[k][h] + rcl[h] % 3, v[k][h]);
        }
        double* out = vals + (size_t)S * S * r0;
        if (overwrite) for (int k = lane; k < S * S * cnt; k += 64) out[k] = acc[k];
        else for (int k = lane; k < S * S * cnt; k += 64) out[k] += acc[k];
    }
}

}  // namespace fenris_hip
