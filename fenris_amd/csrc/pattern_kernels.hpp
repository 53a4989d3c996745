// Integer kernels: node->element adjacency, node-level sparsity pattern, CSR expansion.
// Replaces CsrAssembler::assemble_pattern / CsrParAssembler::assemble_pattern
// (src/assembly/global.rs:65-120, 206-297): the per-node FxHashSet + sort becomes
// "gather candidates of the adjacent elements, bitonic-sort in LDS, unique".
// The output is unique given the connectivity, hence bit-identical to the reference.
#pragma once
#include "device_common.hpp"

namespace fenris_hip {

// connectivity view that covers both fixed-n meshes (eoff == null, nodes[e*n + a]) and ragged lists
struct ConnView {
    const int* nodes;       // flat node list
    const unsigned* eoff;   // E+1 offsets or null
    const unsigned* k2e;    // flat index -> element (ragged only)
    int n;                  // nodes per element when eoff == null
    long long total;        // flat length
    const unsigned char* active;  // optional per-element mask (compute adjacency); null = all elements
    __device__ __forceinline__ void elem_range(unsigned e, unsigned& b, unsigned& en) const {
        if (eoff) { b = eoff[e]; en = eoff[e + 1]; } else { b = e * (unsigned)n; en = b + (unsigned)n; }
    }
    __device__ __forceinline__ unsigned elem_of(unsigned k) const { return eoff ? k2e[k] : k / (unsigned)n; }
};

static __global__ void k_count_degree(ConnView c, unsigned* deg, int num_nodes, int* bad) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < c.total; k += (long long)gridDim.x * blockDim.x) {
        const int node = c.nodes[k];
        if (node < 0 || node >= num_nodes) { *bad = 1; continue; }
        if (c.active && !c.active[c.elem_of((unsigned)k)]) continue;
        atomicAdd(&deg[node], 1u);
    }
}

static __global__ void k_fill_n2e(ConnView c, const unsigned* n2e_off, unsigned* cursor, unsigned* n2e, int num_nodes) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < c.total; k += (long long)gridDim.x * blockDim.x) {
        const int node = c.nodes[k];
        if (node < 0 || node >= num_nodes) continue;
        if (c.active && !c.active[c.elem_of((unsigned)k)]) continue;
        const unsigned p = atomicAdd(&cursor[node], 1u);
        n2e[n2e_off[node] + p] = (unsigned)k;
    }
}

// per node: ascending flat index == ascending (element, local index): deterministic adjacency
static __global__ void k_sort_n2e(const unsigned* n2e_off, unsigned* n2e, int num_nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_nodes) return;
    const unsigned b = n2e_off[i], en = n2e_off[i + 1];
    for (unsigned x = b + 1; x < en; ++x) {
        const unsigned v = n2e[x];
        unsigned y = x;
        while (y > b && n2e[y - 1] > v) { n2e[y] = n2e[y - 1]; --y; }
        n2e[y] = v;
    }
}

// One wave per node: candidates = all nodes of all adjacent elements; sort + unique in LDS.
// FILL == false: cnt[node] = number of distinct neighbours; FILL == true: write them at noff[node].
constexpr int NEIGH_CAP = 4096;
// A node with more than NEIGH_CAP candidates (a hub shared by thousands of elements: nothing a finite element mesh has, but the reference
// walks any connectivity, global.rs:65-120) goes to the list `heavy` in the counting pass and is skipped here; k_heavy_neighbors does it
// with a bitmap over all nodes.
constexpr int HEAVY_CAP = 4096;   // such nodes per pattern
template <bool FILL>
__global__ void __launch_bounds__(64) k_node_neighbors(ConnView c, const unsigned* n2e_off, const unsigned* n2e, int num_nodes,
                                                       unsigned* cnt, const unsigned* noff, unsigned* ncols, int* overflow, int* heavy) {
    __shared__ unsigned buf[NEIGH_CAP];
    const int lane = threadIdx.x;
    // fast path (fixed-n meshes, at most 64 candidates -- e.g. the 8 x 8 of a hexahedral mesh): one candidate per lane.  The loads of a
    // node are three dependent round trips (offsets -> entry -> node id); the candidates of the wavefront's NEXT node are requested
    // before the current node is sorted (no gain measured in this two-pass form: 22.9 -> 24.4 ms per build on 216^3; kept for the one-pass form's sake, which has it too)
    auto fetch = [&](int node, unsigned& b, unsigned& en, unsigned& v) {
        v = 0xffffffffu;
        b = en = 0;
        if (node >= num_nodes) return;
        b = n2e_off[node];
        en = n2e_off[node + 1];
        if (!c.eoff && (en - b) * (unsigned)c.n <= 64u && (unsigned)lane < (en - b) * (unsigned)c.n) {
            const unsigned e = n2e[b + (unsigned)lane / (unsigned)c.n] / (unsigned)c.n;
            v = (unsigned)c.nodes[(size_t)e * c.n + (unsigned)lane % (unsigned)c.n];
        }
    };
    unsigned b_n, en_n, v_n;
    fetch(blockIdx.x, b_n, en_n, v_n);
    for (int node = blockIdx.x; node < num_nodes; node += gridDim.x) {
        const unsigned b = b_n, en = en_n;
        unsigned v = v_n;
        fetch(node + gridDim.x, b_n, en_n, v_n);
        if (!c.eoff && (en - b) * (unsigned)c.n <= 64u) {
            // bitonic sort across the wavefront in registers, no LDS and no barriers.  A degenerate element that
            // lists the node twice only duplicates candidates, which the unique step removes.
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const unsigned other = (unsigned)__shfl_xor((int)v, stride, 64);
                    const bool up = ((lane & size) == 0), lower = ((lane & stride) == 0);
                    v = (lower == up) ? min(v, other) : max(v, other);
                }
            const unsigned prev = (unsigned)__shfl_up((int)v, 1, 64);
            const bool keep = (v != 0xffffffffu) && (lane == 0 || v != prev);
            const unsigned long long mask = __ballot(keep);
            if (FILL) {
                if (keep) ncols[noff[node] + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = v;
            } else if (lane == 0) {
                cnt[node] = (unsigned)__popcll(mask);
            }
            continue;
        }
        // collect candidates; consecutive entries of the same element (degenerate elements) are skipped
        __syncthreads();
        unsigned C = 0;
        unsigned prev_e = 0xffffffffu;
        for (unsigned x = b; x < en; ++x) {  // uniform loop
            const unsigned e = c.elem_of(n2e[x]);
            if (e == prev_e) continue;
            prev_e = e;
            unsigned eb, ee;
            c.elem_range(e, eb, ee);
            for (unsigned k = eb + lane; k < ee; k += 64)
                if (C + (k - eb) < NEIGH_CAP) buf[C + (k - eb)] = (unsigned)c.nodes[k];
            C += ee - eb;
        }
        if (C > NEIGH_CAP) {
            if (lane == 0 && !FILL) {
                const int k = atomicAdd(overflow, 1);
                if (k < HEAVY_CAP) heavy[k] = node;
                cnt[node] = 0;
            }
            __syncthreads();
            continue;
        }
        unsigned P = 64;
        while (P < C) P <<= 1;
        for (unsigned k = C + lane; k < P; k += 64) buf[k] = 0xffffffffu;
        __syncthreads();
        // bitonic sort, ascending
        for (unsigned size = 2; size <= P; size <<= 1)
            for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
                for (unsigned t = lane; t < (P >> 1); t += 64) {
                    const unsigned lo = 2 * t - (t & (stride - 1));
                    const unsigned hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const unsigned x = buf[lo], y = buf[hi];
                    if ((x > y) == up) { buf[lo] = y; buf[hi] = x; }
                }
                __syncthreads();
            }
        // unique: element k is kept iff k == 0 or differs from its predecessor; rank by wave scan
        unsigned base = 0;
        for (unsigned k0 = 0; k0 < C; k0 += 64) {
            const unsigned k = k0 + lane;
            const bool keep = (k < C) && (k == 0 || buf[k] != buf[k - 1]);
            const unsigned long long mask = __ballot(keep);
            if (FILL && keep) {
                const unsigned r = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                ncols[noff[node] + r] = buf[k];
            }
            base += (unsigned)__popcll(mask);
        }
        if (!FILL && lane == 0) cnt[node] = base;
        __syncthreads();
    }
}

// 128 keys across a wavefront, two per lane (key i = 64 w + lane in vw), ascending: the bitonic network of the 64-key form with the
// stride-64 exchange done inside the lane
__device__ __forceinline__ void wave_sort128(unsigned& v0, unsigned& v1, int lane) {
#pragma unroll
    for (int size = 2; size <= 128; size <<= 1)
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride == 64) {   // (size == 128: ascending everywhere)
                const unsigned lo = min(v0, v1), hi = max(v0, v1);
                v0 = lo;
                v1 = hi;
            } else {
                const unsigned o0 = (unsigned)__shfl_xor((int)v0, stride, 64), o1 = (unsigned)__shfl_xor((int)v1, stride, 64);
                const bool lower = ((lane & stride) == 0);
                const bool up0 = size == 128 ? true : ((lane & size) == 0);
                const bool up1 = size == 128 ? true : (((64 + lane) & size) == 0);
                v0 = (lower == up0) ? min(v0, o0) : max(v0, o0);
                v1 = (lower == up1) ? min(v1, o1) : max(v1, o1);
            }
        }
}

// Fixed-n meshes whose nodes all have at most 64 W candidates (W = 1: hexahedral meshes, 8 x 8; W = 2: tetrahedral meshes, ~24 x 4): ONE
// neighbour pass instead of a counting and a filling one.  A wavefront sorts a node's candidates in registers and leaves the distinct
// ones in a scratch row of 64 entries (tmp[node][rank]) together with their number; after the scan of the counts k_compact_neighbors
// moves the rows to their places.  The candidates of the NEXT node of the wavefront are requested before the current node is sorted.
// *overflow is set when a node has more than 64 DISTINCT neighbours (W = 2 only: the caller falls back to the two passes).
template <int W>
__global__ void __launch_bounds__(64) k_node_neighbors_once(const int* nodes, int n, const unsigned* n2e_off, const unsigned* n2e, int num_nodes,
                                                            unsigned* cnt, unsigned* tmp, int* overflow) {
    const int lane = threadIdx.x;
    auto fetch = [&](int node, unsigned (&v)[W]) {
#pragma unroll
        for (int w = 0; w < W; ++w) v[w] = 0xffffffffu;
        if (node >= num_nodes) return;
        const unsigned b = n2e_off[node], C = (n2e_off[node + 1] - b) * (unsigned)n;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const unsigned i = (unsigned)lane + 64u * w;
            if (i < C) {
                const unsigned e = n2e[b + i / (unsigned)n] / (unsigned)n;
                v[w] = (unsigned)nodes[(size_t)e * n + i % (unsigned)n];
            }
        }
    };
    int node = blockIdx.x;
    unsigned v_next[W];
    fetch(node, v_next);
    for (; node < num_nodes; node += gridDim.x) {
        unsigned v[W];
#pragma unroll
        for (int w = 0; w < W; ++w) v[w] = v_next[w];
        fetch(node + gridDim.x, v_next);
        if constexpr (W == 1) {
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const unsigned other = (unsigned)__shfl_xor((int)v[0], stride, 64);
                    const bool up = ((lane & size) == 0), lower = ((lane & stride) == 0);
                    v[0] = (lower == up) ? min(v[0], other) : max(v[0], other);
                }
        } else {
            wave_sort128(v[0], v[W - 1], lane);
        }
        unsigned base = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            unsigned prev = (unsigned)__shfl_up((int)v[w], 1, 64);
            if (w > 0) { const unsigned last = (unsigned)__shfl((int)v[w - 1], 63, 64); if (lane == 0) prev = last; }
            const bool keep = (v[w] != 0xffffffffu) && ((w == 0 && lane == 0) || v[w] != prev);
            const unsigned long long mask = __ballot(keep);
            const unsigned r = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
            if (keep && r < 64u) tmp[(size_t)node * 64 + r] = v[w];
            base += (unsigned)__popcll(mask);
        }
        if (lane == 0) {
            cnt[node] = min(base, 64u);
            if (base > 64u) *overflow = 1;
        }
    }
}
// 256 consecutive nodes per workgroup: their rows are one contiguous piece of ncols, written in order (coalesced); the node of
// an entry by bisection in the workgroup's 257 offsets
static __global__ void __launch_bounds__(256) k_compact_neighbors(const unsigned* noff, const unsigned* tmp, int num_nodes, unsigned* ncols) {
    __shared__ unsigned off[257];
    const int n0 = blockIdx.x * 256, nn = min(256, num_nodes - n0);
    for (int i = threadIdx.x; i <= nn; i += 256) off[i] = noff[n0 + i];
    __syncthreads();
    const unsigned first = off[0], last = off[nn];
    for (unsigned j = first + threadIdx.x; j < last; j += 256) {
        int lo = 0, hi = nn;            // largest i with off[i] <= j
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (off[mid] <= j) lo = mid; else hi = mid;
        }
        ncols[j] = tmp[(size_t)(n0 + lo) * 64 + (j - off[lo])];
    }
}

// The neighbours of the heavy nodes: one workgroup per node at a time, a bitmap over all nodes in global memory (its own slice per
// workgroup), bits set by atomics, then counted (FILL == false) or written out in ascending order (FILL == true).
template <bool FILL>
__global__ void __launch_bounds__(256) k_heavy_neighbors(ConnView c, const unsigned* n2e_off, const unsigned* n2e, const int* heavy, int nheavy,
                                                        int num_nodes, unsigned* bitmap, int W, unsigned* cnt, const unsigned* noff,
                                                        unsigned* ncols) {
    __shared__ unsigned part[256];
    const int tid = threadIdx.x;
    unsigned* bm = bitmap + (size_t)blockIdx.x * W;
    for (int h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const int node = heavy[h];
        for (int w = tid; w < W; w += 256) bm[w] = 0u;
        __syncthreads();
        unsigned prev_e = 0xffffffffu;
        for (unsigned x = n2e_off[node]; x < n2e_off[node + 1]; ++x) {  // uniform loop
            const unsigned e = c.elem_of(n2e[x]);
            if (e == prev_e) continue;
            prev_e = e;
            unsigned eb, ee;
            c.elem_range(e, eb, ee);
            for (unsigned k = eb + tid; k < ee; k += 256) {
                const unsigned v = (unsigned)c.nodes[k];
                // (an index >= num_nodes is reported through flags[0] by the counting pass; the host reads that flag only after
                // this kernel: never mark outside the bitmap)
                if (v < (unsigned)num_nodes) atomicOr(&bm[v >> 5], 1u << (v & 31u));
            }
        }
        __syncthreads();
        const int chunk = (W + 255) / 256, w0 = min(W, tid * chunk), w1 = min(W, w0 + chunk);
        unsigned local = 0;
        for (int w = w0; w < w1; ++w) local += (unsigned)__popc(bm[w]);
        part[tid] = local;
        __syncthreads();
        if (tid == 0) {   // exclusive scan of 256 counts: the node is one in a million
            unsigned run = 0;
            for (int t = 0; t < 256; ++t) { const unsigned v = part[t]; part[t] = run; run += v; }
            if (!FILL) cnt[node] = run;
        }
        __syncthreads();
        if (FILL) {
            unsigned r = noff[node] + part[tid];
            for (int w = w0; w < w1; ++w) {
                unsigned bits = bm[w];
                while (bits) {
                    const int b = __ffs((int)bits) - 1;
                    ncols[r++] = (unsigned)(32 * w + b);
                    bits &= bits - 1u;
                }
            }
        }
        __syncthreads();
    }
}

// scalar CSR row offsets: rows s*i + r  (global.rs:83-93)
static __global__ void k_expand_row_offsets(const unsigned* noff, int num_nodes, int S, unsigned long long* row_offsets) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long R = (long long)num_nodes * S;
    if (t > R) return;
    if (t == R) { row_offsets[R] = (unsigned long long)S * S * noff[num_nodes]; return; }
    const int i = (int)(t / S), r = (int)(t % S);
    const unsigned long long cnt = noff[i + 1] - noff[i];
    row_offsets[t] = (unsigned long long)S * S * noff[i] + (unsigned long long)r * S * cnt;
}

// scalar CSR column indices: s*j + c, sdim identical rows per node (global.rs:97-110)
static __global__ void k_expand_col_indices(const unsigned* noff, const unsigned* ncols, int num_nodes, int S,
                                     unsigned long long* col_indices) {
    const long long nnzn = noff[num_nodes];
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nnzn; t += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = num_nodes;  // last i with noff[i] <= t
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((long long)noff[mid] <= t) lo = mid; else hi = mid;
        }
        const int i = lo;
        const unsigned long long cnt = noff[i + 1] - noff[i];
        const unsigned long long k = (unsigned long long)t - noff[i];
        const unsigned long long j = ncols[t];
        unsigned long long* base = col_indices + (unsigned long long)S * S * noff[i] + (unsigned long long)S * k;
        for (int r = 0; r < S; ++r)
            for (int cc = 0; cc < S; ++cc) base[(unsigned long long)r * S * cnt + cc] = (unsigned long long)S * j + cc;
    }
}

static __global__ void k_narrow_connectivity(const unsigned long long* in, int* out, long long total, int num_nodes, int* bad) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < total; k += (long long)gridDim.x * blockDim.x) {
        const unsigned long long v = in[k];
        if (v >= (unsigned long long)num_nodes) { *bad = 1; out[k] = 0; } else out[k] = (int)v;
    }
}

// apply_homogeneous_dirichlet_bc_csr (global.rs:379-451) on node-level structure:
// pass 1 marks Dirichlet rows/cols, pass 2 rewrites values.  member[] has one byte per node.
// Dirichlet rows and columns (global.rs:379-451): half a wavefront per node row, one lane per column block (rows longer than 32 blocks in
// trips): the entry's row is known from the workgroup's place -- the first form searched the row of each of the nnz_n entries by
// bisection in the offsets (23 dependent loads per entry: 7 of the 11 ms this step took on the 216^3 mesh).
static __global__ void __launch_bounds__(256) k_dirichlet_rows(const unsigned* noff, const unsigned* ncols, int num_nodes, int S, const unsigned char* member,
                                                        double* vals, const double* scale_dev) {
    const int hl = threadIdx.x & 31;
    const double scale = *scale_dev;
    for (long long i = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); i < num_nodes; i += (long long)gridDim.x * 8) {
        const unsigned b = noff[i], cnt = noff[i + 1] - b;
        const bool ri = member[i] != 0;
        for (unsigned k = hl; k < cnt; k += 32) {
            const unsigned j = ncols[b + k];
            const bool cj = member[j] != 0;
            if (!ri && !cj) continue;
            double* base = vals + (unsigned long long)S * S * b + (unsigned long long)S * k;
            for (int r = 0; r < S; ++r)
                for (int c = 0; c < S; ++c)
                    base[(unsigned long long)r * S * cnt + c] = (ri && (unsigned)i == j && r == c) ? scale : 0.0;
        }
    }
}

static __global__ void __launch_bounds__(256) k_mark_nodes(const unsigned long long* nodes, long long n, unsigned char* member) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t < n) member[nodes[t]] = 1;
}

static __global__ void k_dirichlet_rhs(double* rhs, const unsigned long long* nodes, long long n, int S) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= n * S) return;
    rhs[nodes[t / S] * S + (t % S)] = 0.0;
}

// first non-zero diagonal entry in row order (global.rs:387-395).
// pass 1 (out == null): atomicMin of the first scalar row whose diagonal is non-zero;
// pass 2 (out != null): thread 0 writes |diag| of row *first.
__device__ __forceinline__ double diag_value(const unsigned* noff, const unsigned* ncols, int S, const double* vals, long long t) {
    const int i = (int)(t / S), r = (int)(t % S);
    const unsigned b = noff[i], cnt = noff[i + 1] - b;
    for (unsigned k = 0; k < cnt; ++k)
        if (ncols[b + k] == (unsigned)i)
            return vals[(unsigned long long)S * S * b + (unsigned long long)r * S * cnt + (unsigned long long)S * k + r];
    return 0.0;
}
static __global__ void k_first_nonzero_diag(const unsigned* noff, const unsigned* ncols, int num_nodes, int S, const double* vals,
                                     unsigned long long* first, double* out) {
    if (out) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            *out = (*first == ~0ull) ? 1.0 : fabs(diag_value(noff, ncols, S, vals, (long long)*first));
        return;
    }
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= (long long)num_nodes * S) return;
    // (an earlier row has already answered -- normally row 0 -- : no need to look at this one)
    if ((unsigned long long)t > *reinterpret_cast<volatile unsigned long long*>(first)) return;
    if (diag_value(noff, ncols, S, vals, t) != 0.0) atomicMin(first, (unsigned long long)t);
}

}  // namespace fenris_hip
