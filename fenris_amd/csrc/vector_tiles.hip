// Residual vector through element TILES (round 4): the element vectors never reach HBM.
//
// Round 3's two passes (element_pass.hpp) wrote every element vector to scratch (Hex8 elasticity: 192 bytes per element) and read it back
// per node: 1.9 GB each way on Hex8 216^3 for 1.06 GB of algorithmic traffic.  Here the elements are cut into TILES of 256 that are
// compact in space (consecutive runs of the Morton order of the element centroids: 8 x 8 x 4 bricks on a structured mesh, clusters on
// any other), one workgroup per tile:
//
//  * every thread computes its element's vector in registers exactly as before (element_pass_body: assemble_element_elliptic_vector,
//    src/assembly/local/elliptic.rs:457-531) and leaves it in LDS, stage[a][c][thread];
//  * the tile's DISTINCT nodes (~405 of a brick's 2048 (element, local node) entries) are summed from LDS, one thread per node over its
//    entries in ascending (element, local node) order, and only these partial sums go to HBM: partial[P][S], ~1.6 per element
//    instead of 8 (38 instead of 192 bytes per Hex8 element);
//  * the node pass (k_vector_from_partials) adds the partials of every node in ascending tile order (1.6 on average) to the output:
//    the work of VectorAssembler::assemble_vector_into's add_local_to_global (global.rs:582-608, 770-796) without atomics, in an order
//    fixed by the tables: bitwise reproducible.
//
// The tables (tile membership, local node numbers, in-tile adjacency, node -> partials) depend on the connectivity only (the vertex
// coordinates merely decide how compact the tiles are) and are built on the device: Morton keys, hipcub radix sort, one workgroup
// per tile that sorts its 2048 (node, entry) keys in LDS, two scans.  Element masks (fh_set_active_elements: the multi-GPU partitions)
// zero the inactive elements' contributions.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "element_pass.hpp"
#include "vector_tiles.hpp"

namespace fenris_hip {

namespace {

constexpr int VT_TS = 256;   // elements per tile = threads per workgroup of the element pass (the kernels take it as a template parameter)

struct Box3 { double lo[3], scale[3]; };

__global__ void __launch_bounds__(256) k_vt_bbox(const double* verts, int num_nodes, int D, double* out /* [grid][6] */) {
    __shared__ double red[6][256];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = blockIdx.x * 256 + threadIdx.x; v < num_nodes; v += gridDim.x * 256)
        for (int i = 0; i < D; ++i) {
            const double x = verts[(size_t)v * D + i];
            if (x < lo[i]) lo[i] = x;     // (NaN coordinates compare false: ignored)
            if (x > hi[i]) hi[i] = x;
        }
    for (int i = 0; i < 3; ++i) { red[i][threadIdx.x] = lo[i]; red[3 + i][threadIdx.x] = hi[i]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int i = 0; i < 3; ++i) {
                red[i][threadIdx.x] = fmin(red[i][threadIdx.x], red[i][threadIdx.x + s]);
                red[3 + i][threadIdx.x] = fmax(red[3 + i][threadIdx.x], red[3 + i][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) out[(size_t)blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__device__ __forceinline__ unsigned long long vt_spread21(unsigned long long v) {   // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ void __launch_bounds__(256) k_vt_morton(const int* conn, int n, long long E, const double* verts, int D, int num_nodes, Box3 box,
                                                   unsigned long long* keys, int* vals, int* bad) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    double c[3] = {0.0, 0.0, 0.0};
    for (int a = 0; a < n; ++a) {
        const int v = conn[(size_t)e * n + a];
        if (v < 0 || v >= num_nodes) { *bad = 1; continue; }
        for (int i = 0; i < D; ++i) c[i] += verts[(size_t)v * D + i];
    }
    unsigned long long key = 0;
    for (int i = 0; i < D; ++i) {
        double q = (c[i] / n - box.lo[i]) * box.scale[i];
        if (!(q >= 0.0)) q = 0.0;          // also NaN
        if (q > 2097151.0) q = 2097151.0;
        key |= vt_spread21((unsigned long long)q) << i;
    }
    keys[e] = key;
    vals[e] = (int)e;
}

template <typename T>
__device__ __forceinline__ void vt_bitonic_sort(T* d, int M, int nthreads) {   // ascending, M a power of two
    for (int k = 2; k <= M; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < M; i += nthreads) {
                const int o = i ^ j;
                if (o > i) {
                    const T x = d[i], y = d[o];
                    if ((x > y) == ((i & k) == 0)) { d[i] = y; d[o] = x; }
                }
            }
            __syncthreads();
        }
}

constexpr unsigned long long VT_PAD = ~0ull;

// one workgroup per tile.  FILL = false: number of distinct nodes of the tile;  FILL = true: all tables of the tile
template <bool FILL, int TS>
__global__ void __launch_bounds__(TS) k_vt_tile_tables(const int* order, long long E, const int* conn, int n, unsigned* tile_cnt, const unsigned* noff,
                                                        int* elem, int* tconn, unsigned short* la_off, unsigned short* la, unsigned* p_node) {   // p_node: global node of every partial
    __shared__ unsigned long long key[TS * 8];
    __shared__ unsigned scan[TS];
    const int tile = blockIdx.x, t = threadIdx.x;
    const long long idx = (long long)tile * TS + t;
    key[t] = idx < E ? (unsigned long long)(unsigned)order[idx] : VT_PAD;
    __syncthreads();
    vt_bitonic_sort(key, TS, TS);                                  // elements ascending inside the tile
    const long long el = key[t] == VT_PAD ? -1 : (long long)key[t];
    __syncthreads();
    if (FILL) elem[idx] = (int)el;
    const int M = TS * n;                                       // <= 2048 entries thread n + a
    // (threads without an element read the tile's first element: the kernel computes on every lane and drops their results)
    const long long er = el >= 0 ? el : (long long)key[0];
    for (int a = 0; a < n; ++a) {
        const int node = conn[(size_t)er * n + a];
        if (FILL) tconn[((size_t)tile * n + a) * TS + t] = node;
    }
    __syncthreads();
    for (int i = t; i < TS * 8; i += TS) key[i] = VT_PAD;
    __syncthreads();
    if (el >= 0)
        for (int a = 0; a < n; ++a) key[t * n + a] = ((unsigned long long)(unsigned)conn[(size_t)el * n + a] << 11) | (unsigned)(t * n + a);
    __syncthreads();
    vt_bitonic_sort(key, TS * 8, TS);                                 // by node, then by entry; the padding last
    // heads of the runs of equal nodes: eight consecutive places per thread, then a scan over the threads
    unsigned heads = 0, nvalid = 0;
    for (int j = 0; j < 8; ++j) {
        const int i = t * 8 + j;
        const unsigned long long k = key[i];
        if (k != VT_PAD) {
            ++nvalid;
            if (i == 0 || (key[i - 1] >> 11) != (k >> 11)) ++heads;
        }
    }
    scan[t] = heads;
    __syncthreads();
    for (int s = 1; s < TS; s <<= 1) {
        const unsigned v = t >= s ? scan[t - s] : 0u;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const unsigned U = scan[TS - 1];
    if (!FILL) {
        if (t == 0) tile_cnt[tile] = U;
        return;
    }
    // number of valid entries = M minus the padding of absent elements: count through a second use of the scan array
    unsigned ln = scan[t] - heads;                               // distinct nodes before this thread's places
    __syncthreads();
    scan[t] = nvalid;
    __syncthreads();
    for (int s = TS / 2; s > 0; s >>= 1) {
        if (t < s) scan[t] += scan[t + s];
        __syncthreads();
    }
    const unsigned total = scan[0];
    const unsigned base = noff[tile];
    unsigned short* lo = la_off + base + tile;
    unsigned short* lat = la + (size_t)tile * M;
    for (int j = 0; j < 8; ++j) {
        const int i = t * 8 + j;
        const unsigned long long k = key[i];
        if (k == VT_PAD) continue;
        const bool head = i == 0 || (key[i - 1] >> 11) != (k >> 11);
        if (head) {
            lo[ln] = (unsigned short)i;
            p_node[base + ln] = (unsigned)(k >> 11);
            ++ln;
        }
        const unsigned ent = (unsigned)(k & 2047u);
        lat[i] = (unsigned short)ent;
    }
    if (t == 0) lo[U] = (unsigned short)total;
}

__global__ void __launch_bounds__(256) k_vt_count(const unsigned* p_node, unsigned P, unsigned* deg) {
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if (p < P) atomicAdd(&deg[p_node[p]], 1u);
}
__global__ void __launch_bounds__(256) k_vt_fill(const unsigned* p_node, unsigned P, const unsigned* np_off, unsigned* cursor, unsigned* np_idx) {
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if (p < P) {
        const unsigned v = p_node[p];
        np_idx[np_off[v] + atomicAdd(&cursor[v], 1u)] = p;
    }
}
__global__ void __launch_bounds__(256) k_vt_sort_lists(const unsigned* np_off, unsigned* np_idx, int num_nodes) {   // short lists: insertion sort
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= num_nodes) return;
    const unsigned k0 = np_off[v], k1 = np_off[v + 1];
    for (unsigned i = k0 + 1; i < k1; ++i) {
        const unsigned x = np_idx[i];
        unsigned j = i;
        while (j > k0 && np_idx[j - 1] > x) { np_idx[j] = np_idx[j - 1]; --j; }
        np_idx[j] = x;
    }
}

// ---- the element pass over tiles
// tile of workgroup b when the grid is 8 ceil(T / 8) workgroups: XCD b mod 8 walks the tiles [x c, (x + 1) c), c = ceil(T / 8)
__device__ __forceinline__ int xcd_tile(int b, int ntiles) {
    const int c = (ntiles + 7) >> 3;
    return (b & 7) * c + (b >> 3);
}
// what a tile's workgroup needs for the node sums, requested before the arithmetic and landing behind it: the tile's entries (256 N
// sixteen-bit values: N / 4 eight-byte pieces per thread) and the starts of the first 512 distinct nodes
template <int N, int TS>
struct TileSums {
    static constexpr int EW = (N * 2 + 7) / 8;
    unsigned base, U;
    const unsigned short* lo;
    uint2 ent_w[EW];
    unsigned short st0[2], st1[2];
    __device__ __forceinline__ void request(const VecTiles& t, int tile, int tid) {
        base = t.noff[tile];
        U = t.noff[tile + 1] - base;
        lo = t.la_off + base + tile;
        const uint2* la8 = reinterpret_cast<const uint2*>(t.la + (size_t)tile * (TS * N));   // (TS N x 2 bytes per tile: 8-byte aligned for even N)
        if constexpr (N % 4 == 0) {
#pragma unroll
            for (int w = 0; w < EW; ++w) ent_w[w] = la8[w * TS + tid];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned p = tid + TS * j;
            st0[j] = st1[j] = 0;
            if (p < U) { st0[j] = lo[p]; st1[j] = lo[p + 1]; }
        }
    }
    // stage[a][c][thread] holds the element vectors (SV components); ents: LDS place for the entries.  Ends with the partials stored.
    template <int SV>
    __device__ __forceinline__ void sum(const VecTiles& t, int tile, int tid, const double* stage, unsigned short* ents, double* partial) {
        if constexpr (N % 4 == 0) {
            uint2* e8 = reinterpret_cast<uint2*>(ents);
#pragma unroll
            for (int w = 0; w < EW; ++w) e8[w * TS + tid] = ent_w[w];
        }
        __syncthreads();
        const unsigned short* la = t.la + (size_t)tile * (TS * N);
        auto sum_node = [&](unsigned p, unsigned k0, unsigned k1) {
            double acc[SV];
#pragma unroll
            for (int c = 0; c < SV; ++c) acc[c] = 0.0;
            for (unsigned k = k0; k < k1; ++k) {
                const unsigned ent = (N % 4 == 0) ? ents[k] : la[k], tt = ent / (unsigned)N, aa = ent - tt * (unsigned)N;
#pragma unroll
                for (int c = 0; c < SV; ++c) acc[c] += stage[(aa * SV + c) * TS + tt];
            }
#pragma unroll
            for (int c = 0; c < SV; ++c) partial[(size_t)(base + p) * SV + c] = acc[c];
        };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned p = tid + TS * j;
            if (p < U) sum_node(p, st0[j], st1[j]);
        }
        for (unsigned p = tid + 2 * TS; p < U; p += TS) sum_node(p, lo[p], lo[p + 1]);   // (tiles of scattered elements)
    }
};

template <int EK, int OP, int TS, int MONO = 0>
__global__ void __launch_bounds__(TS) k_element_pass_tiled(const KArgs a, const VecTiles t, const unsigned char* active, double* partial) {
    constexpr int N = EPDims<EK, OP, EP_VECTOR>::N, S = EPDims<EK, OP, EP_VECTOR>::S, D = EPDims<EK, OP, EP_VECTOR>::D;
    __shared__ double stage[N * S * TS];
    __shared__ unsigned short ents[(N % 4 == 0) ? TS * N : 4];
    // workgroup b runs on XCD b mod 8 (round-robin dispatch): every XCD takes one contiguous eighth of the tiles, so that tiles that are
    // neighbours in space (consecutive in the Morton order) share their boundary nodes' coordinates and u through one L2
    const int tile = xcd_tile((int)blockIdx.x, t.ntiles), tid = threadIdx.x;
    if (tile >= t.ntiles) return;
    const int el = t.elem[(size_t)tile * TS + tid];
    const bool live = el >= 0 && (!active || active[el] != 0);
    const long long ec = el >= 0 ? el : 0;
    TileSums<N, TS> ts;
    ts.request(t, tile, tid);
    // the tile's connectivity comes from a table laid out by tile thread (tconn[tile][a][thread]: coalesced, and not behind the load
    // of the element id)
    double X[N][D], Uv[N][S];
    {
        int nd[N];
#pragma unroll
        for (int n = 0; n < N; ++n) nd[n] = t.tconn[((size_t)tile * N + n) * TS + tid];
        if (a.ablate & 128) {   // (profiling: what the scattered gathers cost -- every thread reads the nodes of its wavefront's first element: wrong results)
#pragma unroll
            for (int n = 0; n < N; ++n) nd[n] = __builtin_amdgcn_readfirstlane(nd[n]);
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
#pragma unroll
            for (int i = 0; i < D; ++i) X[n][i] = a.verts[(size_t)nd[n] * D + i];
#pragma unroll
            for (int k = 0; k < S; ++k) Uv[n][k] = a.u ? a.u[(size_t)nd[n] * S + k] : 0.0;
        }
    }
    // (measured and dropped, Hex8 elasticity 216^3, element pass 1.03 ms: the distinct nodes' coordinates and u through LDS instead of
    // the gathers -- two more barriers, nothing in flight across them: 1.35 ms; u parked in LDS and 168 registers for a third
    // wavefront per SIMD -- 62 registers spill: 1.66 ms)
    double f[N][S];
    double energy;
    element_pass_body<EK, OP, EP_VECTOR, MONO>(a, el, live, ec, X, EPRegU<N, S>{Uv}, f, energy);
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int c = 0; c < S; ++c) stage[(n * S + c) * TS + tid] = live ? f[n][c] : 0.0;   // (select: a skipped element may hold NaN)
    if (a.ablate & 64) return;   // (profiling: the element phase alone)
    ts.template sum<S>(t, tile, tid, stage, ents, partial);
}

// energy (compute_element_elliptic_energy, local/elliptic.rs:551-605) over the same tiles: on a numbering without locality the tile order
// is what makes the gathers of the coordinates and of u local; the tile's elements are summed in a fixed tree, one partial per workgroup
// (workgroups beyond the last tile write a zero), k_sum_partials adds them in index order
template <int EK, int OP, int TS, int MONO = 0>
__global__ void __launch_bounds__(TS) k_element_energy_tiled(const KArgs a, const VecTiles t, const unsigned char* active, double* partial) {
    constexpr int N = EPDims<EK, OP, EP_SCALAR>::N, S = EPDims<EK, OP, EP_SCALAR>::S, D = EPDims<EK, OP, EP_SCALAR>::D;
    static_assert(TS == 256, "block_sum_256");
    __shared__ double red[4];
    const int tile = xcd_tile((int)blockIdx.x, t.ntiles), tid = threadIdx.x;
    if (tile >= t.ntiles) {
        if (tid == 0) partial[blockIdx.x] = 0.0;
        return;
    }
    const int el = t.elem[(size_t)tile * TS + tid];
    const bool live = el >= 0 && (!active || active[el] != 0);
    double X[N][D], Uv[N][S];
    {
        int nd[N];
#pragma unroll
        for (int n = 0; n < N; ++n) nd[n] = t.tconn[((size_t)tile * N + n) * TS + tid];
#pragma unroll
        for (int n = 0; n < N; ++n) {
#pragma unroll
            for (int i = 0; i < D; ++i) X[n][i] = a.verts[(size_t)nd[n] * D + i];
#pragma unroll
            for (int k = 0; k < S; ++k) Uv[n][k] = a.u ? a.u[(size_t)nd[n] * S + k] : 0.0;
        }
    }
    double f[1][S];
    double energy;
    element_pass_body<EK, OP, EP_SCALAR, MONO>(a, el, live, el >= 0 ? el : 0, X, EPRegU<N, S>{Uv}, f, energy);
    const double tot = block_sum_256(live ? energy : 0.0, red);
    if (tid == 0) partial[blockIdx.x] = tot;
}

// source vector (local/source.rs:159-278) over the same tiles: source_element_body per thread, the tile's node sums, partials
template <int D, int S, int N, bool FACT, int TS>
__global__ void __launch_bounds__(TS) k_source_elements_tiled(const KArgs a, const SourceG g, const double* values, const VecTiles t,
                                                               const unsigned char* active, double* partial) {
    constexpr int SF = FACT ? 1 : S;
    __shared__ double stage[N * SF * TS];
    __shared__ unsigned short ents[(N % 4 == 0) ? TS * N : 4];
    const int tile = xcd_tile((int)blockIdx.x, t.ntiles), tid = threadIdx.x;
    if (tile >= t.ntiles) return;
    const int el = t.elem[(size_t)tile * TS + tid];
    const bool live = el >= 0 && (!active || active[el] != 0);
    TileSums<N, TS> ts;
    ts.request(t, tile, tid);
    double X[N][D];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int nd = t.tconn[((size_t)tile * N + n) * TS + tid];
#pragma unroll
        for (int i = 0; i < D; ++i) X[n][i] = a.verts[(size_t)nd * D + i];
    }
    double f[N][SF];
    source_element_body<D, S, N, FACT>(a, g, values, el >= 0 ? el : 0, X, f);
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int c = 0; c < SF; ++c) stage[(n * SF + c) * TS + tid] = live ? f[n][c] : 0.0;
    ts.template sum<SF>(t, tile, tid, stage, ents, partial);
}

// SO > 0: the partials are scalars (S = 1) and the node's SO components are  g[c] sum  (k_source_elements_tiled<FACT>)
template <int S, int SO = 0>
__global__ void __launch_bounds__(256) k_vector_from_partials(int num_nodes, const unsigned* np_off, const unsigned* np_idx, const double* partial,
                                                              double* out, const SourceG g = SourceG{{0.0, 0.0, 0.0}}) {
    static_assert(SO == 0 || S == 1, "scaled node sum: scalar partials");
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= num_nodes) return;
    constexpr int SP = SO > 0 ? SO : S;
    double acc[S], prev[SP];
#pragma unroll
    for (int c = 0; c < S; ++c) acc[c] = 0.0;
#pragma unroll
    for (int c = 0; c < SP; ++c) prev[c] = out[(size_t)node * SP + c];
    const unsigned k0 = np_off[node], k1 = np_off[node + 1];
    for (unsigned kb = k0; kb < k1; kb += 4) {     // four partials requested before the first sum; the additions stay in order
        unsigned v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = np_idx[min(kb + j, k1 - 1)];
        double x[4][S];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < S; ++c) x[j][c] = partial[(size_t)v[j] * S + c];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (kb + j < k1) {
#pragma unroll
                for (int c = 0; c < S; ++c) acc[c] += x[j][c];
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

template <typename T>
hipError_t vt_alloc(VecTilesStore* st, int slot, T** p, size_t count) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (count ? count : 1));
    if (e == hipSuccess) st->bufs[slot] = *p;
    return e;
}

}  // namespace

#define VT_TRY(expr)                                  \
    do {                                              \
        const hipError_t vt_e_ = (expr);              \
        if (vt_e_ != hipSuccess) { cleanup(); return vt_e_; } \
    } while (0)

void VecTilesStore::release() {
    for (void*& b : bufs) {
        if (b) (void)hipFree(b);
        b = nullptr;
    }
    v = VecTiles{};
}

hipError_t vector_tiles_build(hipStream_t stream, const int* conn, int n, long long E, const double* verts, int D, int num_nodes, VecTilesStore* out,
                              int* bad) {
    int ts = VT_TS;
    *bad = 0;
    out->release();
    // (the partial sums are counted in 32 bits: at most E n of them)
    if (E <= 0 || n < 1 || n > 8 || D < 1 || D > 3 || num_nodes <= 0 || E > 0x7fffff00ll || (unsigned long long)E * (unsigned)n >= (1ull << 32) - 256) { *bad = 1; return hipSuccess; }
    // (128-element tiles, two wavefronts per workgroup and four workgroups per CU, were measured: 1.33 against 1.28 ms for the residual of Hex8 216^3)
    const int T = (int)((E + ts - 1) / ts);
    void* tmp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    auto cleanup = [&]() {
        for (void*& p : tmp) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
    };
    // bounding box of the vertices -> Morton keys of the element centroids
    const int bgrid = 256;
    double* d_box = nullptr;
    VT_TRY(hipMalloc(&tmp[0], sizeof(double) * 6 * bgrid));
    d_box = (double*)tmp[0];
    hipLaunchKernelGGL(k_vt_bbox, dim3(bgrid), dim3(256), 0, stream, verts, num_nodes, D, d_box);
    std::vector<double> h_box(6 * bgrid);
    VT_TRY(hipMemcpyAsync(h_box.data(), d_box, sizeof(double) * 6 * bgrid, hipMemcpyDeviceToHost, stream));
    VT_TRY(hipStreamSynchronize(stream));
    Box3 box;
    for (int i = 0; i < 3; ++i) {
        double lo = 1e300, hi = -1e300;
        for (int b = 0; b < bgrid; ++b) { lo = std::min(lo, h_box[6 * b + i]); hi = std::max(hi, h_box[6 * b + 3 + i]); }
        box.lo[i] = lo;
        box.scale[i] = hi > lo ? 2097151.0 / (hi - lo) : 0.0;
    }
    unsigned long long *k_in = nullptr, *k_out = nullptr;
    int *v_in = nullptr, *v_out = nullptr, *d_bad = nullptr;
    VT_TRY(hipMalloc(&tmp[1], sizeof(unsigned long long) * E)); k_in = (unsigned long long*)tmp[1];
    VT_TRY(hipMalloc(&tmp[2], sizeof(unsigned long long) * E)); k_out = (unsigned long long*)tmp[2];
    VT_TRY(hipMalloc(&tmp[3], sizeof(int) * E)); v_in = (int*)tmp[3];
    VT_TRY(hipMalloc(&tmp[4], sizeof(int) * E)); v_out = (int*)tmp[4];
    VT_TRY(hipMalloc(&tmp[5], sizeof(int) * 4)); d_bad = (int*)tmp[5];
    VT_TRY(hipMemsetAsync(d_bad, 0, sizeof(int) * 4, stream));
    hipLaunchKernelGGL(k_vt_morton, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, conn, n, E, verts, D, num_nodes, box, k_in, v_in, d_bad);
    size_t sort_bytes = 0;
    VT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k_in, k_out, v_in, v_out, (int)E, 0, 63, stream));
    VT_TRY(hipMalloc(&tmp[6], sort_bytes + 16));
    VT_TRY(hipcub::DeviceRadixSort::SortPairs(tmp[6], sort_bytes, k_in, k_out, v_in, v_out, (int)E, 0, 63, stream));
    int h_bad = 0;
    VT_TRY(hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream));
    VT_TRY(hipStreamSynchronize(stream));
    if (h_bad) { cleanup(); *bad = 1; return hipSuccess; }
    // keys are done: their buffers hold the tile counts and offsets
    unsigned* tile_cnt = (unsigned*)k_in;     // T + 1 <= E + 1 entries of 4 bytes fit (E >= 1: 8 bytes per key)
    int* elem = nullptr;
    unsigned* noff = nullptr;
    if (vt_alloc(out, 0, &elem, (size_t)T * ts) != hipSuccess || vt_alloc(out, 1, &noff, (size_t)T + 1) != hipSuccess) { cleanup(); out->release(); return hipErrorOutOfMemory; }
    if ((size_t)(T + 1) * sizeof(unsigned) > sizeof(unsigned long long) * (size_t)E) {   // (tiny meshes)
        (void)hipFree(tmp[1]); tmp[1] = nullptr;
        VT_TRY(hipMalloc(&tmp[1], sizeof(unsigned) * ((size_t)T + 1)));
        tile_cnt = (unsigned*)tmp[1];
    }
    VT_TRY(hipMemsetAsync(tile_cnt, 0, sizeof(unsigned) * ((size_t)T + 1), stream));
    hipLaunchKernelGGL((k_vt_tile_tables<false, VT_TS>), dim3(T), dim3(VT_TS), 0, stream, v_out, E, conn, n, tile_cnt, (const unsigned*)nullptr, (int*)nullptr,
                       (int*)nullptr, (unsigned short*)nullptr, (unsigned short*)nullptr, (unsigned*)nullptr);
    size_t scan_bytes = 0;
    VT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, tile_cnt, noff, T + 1, stream));
    if (scan_bytes + 16 > sort_bytes + 16) {
        (void)hipFree(tmp[6]); tmp[6] = nullptr;
        VT_TRY(hipMalloc(&tmp[6], scan_bytes + 16));
    }
    VT_TRY(hipcub::DeviceScan::ExclusiveSum(tmp[6], scan_bytes, tile_cnt, noff, T + 1, stream));
    unsigned P = 0;
    VT_TRY(hipMemcpyAsync(&P, noff + T, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    VT_TRY(hipStreamSynchronize(stream));
    // (more than 2^32 partials would have wrapped the scan: E n < 2^32 is checked by the caller's flat_len; P <= E n)
    unsigned short *la_off = nullptr, *la = nullptr;
    int* tconn = nullptr;
    unsigned *np_off = nullptr, *np_idx = nullptr, *p_node = nullptr;
    hipError_t e = vt_alloc(out, 2, &tconn, (size_t)T * n * ts);
    if (e == hipSuccess) e = vt_alloc(out, 3, &la_off, (size_t)P + T + 1);
    if (e == hipSuccess) e = vt_alloc(out, 4, &la, (size_t)T * n * ts);
    if (e == hipSuccess) e = vt_alloc(out, 5, &np_off, (size_t)num_nodes + 1);
    if (e == hipSuccess) e = vt_alloc(out, 6, &np_idx, (size_t)P);
    if (e != hipSuccess) { cleanup(); out->release(); return e; }
    if (vt_alloc(out, 7, &p_node, (size_t)P + 1) != hipSuccess) { cleanup(); out->release(); return hipErrorOutOfMemory; }
    (void)hipFree(tmp[2]); tmp[2] = nullptr;
    hipLaunchKernelGGL((k_vt_tile_tables<true, VT_TS>), dim3(T), dim3(VT_TS), 0, stream, v_out, E, conn, n, tile_cnt, (const unsigned*)noff, elem, tconn, la_off, la, p_node);
    // node -> partials
    unsigned *deg = nullptr, *cursor = nullptr;
    VT_TRY(hipMalloc(&tmp[7], sizeof(unsigned) * 2 * ((size_t)num_nodes + 1)));
    deg = (unsigned*)tmp[7];
    cursor = deg + num_nodes + 1;
    VT_TRY(hipMemsetAsync(deg, 0, sizeof(unsigned) * 2 * ((size_t)num_nodes + 1), stream));
    if (P) hipLaunchKernelGGL(k_vt_count, dim3((P + 255) / 256), dim3(256), 0, stream, p_node, P, deg);
    VT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, deg, np_off, num_nodes + 1, stream));
    (void)hipFree(tmp[6]); tmp[6] = nullptr;
    VT_TRY(hipMalloc(&tmp[6], scan_bytes + 16));
    VT_TRY(hipcub::DeviceScan::ExclusiveSum(tmp[6], scan_bytes, deg, np_off, num_nodes + 1, stream));
    if (P) {
        hipLaunchKernelGGL(k_vt_fill, dim3((P + 255) / 256), dim3(256), 0, stream, p_node, P, np_off, cursor, np_idx);
        hipLaunchKernelGGL(k_vt_sort_lists, dim3((num_nodes + 255) / 256), dim3(256), 0, stream, np_off, np_idx, num_nodes);
    }
    VT_TRY(hipStreamSynchronize(stream));
    VT_TRY(hipGetLastError());
    cleanup();
    out->v = VecTiles{elem, tconn, noff, la_off, la, np_off, np_idx, p_node, T, n, P, ts};
    return hipSuccess;
}

int vector_tiles_element_pass(int elem_kind, int op, hipStream_t stream, const KArgs& a, const VecTiles& t, const unsigned char* active, double* partial) {
    int rs = -1;
#define VT_OP(EKC)                                                                                                                                   \
    switch (op) {                                                                                                                                    \
        case FH_LAPLACE: hipLaunchKernelGGL((k_element_pass_tiled<EKC, FH_LAPLACE, VT_TS>), dim3(8 * ((t.ntiles + 7) / 8)), dim3(VT_TS), 0, stream, a, t, active, partial); rs = 0; break; \
        case FH_LINEAR_ELASTIC: hipLaunchKernelGGL((k_element_pass_tiled<EKC, FH_LINEAR_ELASTIC, VT_TS>), dim3(8 * ((t.ntiles + 7) / 8)), dim3(VT_TS), 0, stream, a, t, active, partial); rs = 0; break; \
        case FH_NEO_HOOKEAN: hipLaunchKernelGGL((k_element_pass_tiled<EKC, FH_NEO_HOOKEAN, VT_TS>), dim3(8 * ((t.ntiles + 7) / 8)), dim3(VT_TS), 0, stream, a, t, active, partial); rs = 0; break; \
        case FH_STVK: hipLaunchKernelGGL((k_element_pass_tiled<EKC, FH_STVK, VT_TS>), dim3(8 * ((t.ntiles + 7) / 8)), dim3(VT_TS), 0, stream, a, t, active, partial); rs = 0; break; \
        default: break;                                                                                                                              \
    }
    if (elem_kind == FH_HEX8 && a.qmono) {   // the monomial form (element_pass.hpp, round 5)
        const dim3 g(8 * ((t.ntiles + 7) / 8));
        switch (op) {
            case FH_LAPLACE: if (a.all_affine && a.qmom) hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_LAPLACE, VT_TS, 3>), g, dim3(VT_TS), 0, stream, a, t, active, partial); else if (a.all_affine) hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_LAPLACE, VT_TS, 2>), g, dim3(VT_TS), 0, stream, a, t, active, partial); else hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_LAPLACE, VT_TS, 1>), g, dim3(VT_TS), 0, stream, a, t, active, partial); return 0;
            case FH_LINEAR_ELASTIC: if (a.all_affine && a.qmom) hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_LINEAR_ELASTIC, VT_TS, 3>), g, dim3(VT_TS), 0, stream, a, t, active, partial); else if (a.all_affine) hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_LINEAR_ELASTIC, VT_TS, 2>), g, dim3(VT_TS), 0, stream, a, t, active, partial); else hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_LINEAR_ELASTIC, VT_TS, 1>), g, dim3(VT_TS), 0, stream, a, t, active, partial); return 0;
            case FH_NEO_HOOKEAN: if (a.all_affine) hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_NEO_HOOKEAN, VT_TS, 2>), g, dim3(VT_TS), 0, stream, a, t, active, partial); else hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_NEO_HOOKEAN, VT_TS, 1>), g, dim3(VT_TS), 0, stream, a, t, active, partial); return 0;
            case FH_STVK: if (a.all_affine) hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_STVK, VT_TS, 2>), g, dim3(VT_TS), 0, stream, a, t, active, partial); else hipLaunchKernelGGL((k_element_pass_tiled<FH_HEX8, FH_STVK, VT_TS, 1>), g, dim3(VT_TS), 0, stream, a, t, active, partial); return 0;
            default: return -1;
        }
    }
    switch (elem_kind) {
        case FH_QUAD4: VT_OP(FH_QUAD4) break;
        case FH_TRI3: VT_OP(FH_TRI3) break;
        case FH_TET4: VT_OP(FH_TET4) break;
        case FH_HEX8: VT_OP(FH_HEX8) break;
        default: break;
    }
#undef VT_OP
    return rs;
}

int vector_tiles_energy_pass(int elem_kind, int op, hipStream_t stream, const KArgs& a, const VecTiles& t, const unsigned char* active, double* partial) {
    int rs = -1;
    const int grid = 8 * ((t.ntiles + 7) / 8);
#define VT_EN(EKC)                                                                                                                                   \
    switch (op) {                                                                                                                                    \
        case FH_LAPLACE: hipLaunchKernelGGL((k_element_energy_tiled<EKC, FH_LAPLACE, VT_TS>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); rs = grid; break; \
        case FH_LINEAR_ELASTIC: hipLaunchKernelGGL((k_element_energy_tiled<EKC, FH_LINEAR_ELASTIC, VT_TS>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); rs = grid; break; \
        case FH_NEO_HOOKEAN: hipLaunchKernelGGL((k_element_energy_tiled<EKC, FH_NEO_HOOKEAN, VT_TS>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); rs = grid; break; \
        case FH_STVK: hipLaunchKernelGGL((k_element_energy_tiled<EKC, FH_STVK, VT_TS>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); rs = grid; break; \
        default: break;                                                                                                                              \
    }
    if (elem_kind == FH_HEX8 && a.qmono) {
        switch (op) {
            case FH_LAPLACE: if (a.all_affine && a.qmom) hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_LAPLACE, VT_TS, 3>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); else if (a.all_affine) hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_LAPLACE, VT_TS, 2>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); else hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_LAPLACE, VT_TS, 1>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); return grid;
            case FH_LINEAR_ELASTIC: if (a.all_affine && a.qmom) hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_LINEAR_ELASTIC, VT_TS, 3>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); else if (a.all_affine) hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_LINEAR_ELASTIC, VT_TS, 2>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); else hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_LINEAR_ELASTIC, VT_TS, 1>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); return grid;
            case FH_NEO_HOOKEAN: if (a.all_affine) hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_NEO_HOOKEAN, VT_TS, 2>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); else hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_NEO_HOOKEAN, VT_TS, 1>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); return grid;
            case FH_STVK: if (a.all_affine) hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_STVK, VT_TS, 2>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); else hipLaunchKernelGGL((k_element_energy_tiled<FH_HEX8, FH_STVK, VT_TS, 1>), dim3(grid), dim3(VT_TS), 0, stream, a, t, active, partial); return grid;
            default: return -1;
        }
    }
    switch (elem_kind) {
        case FH_QUAD4: VT_EN(FH_QUAD4) break;
        case FH_TRI3: VT_EN(FH_TRI3) break;
        case FH_TET4: VT_EN(FH_TET4) break;
        case FH_HEX8: VT_EN(FH_HEX8) break;
        default: break;
    }
#undef VT_EN
    return rs;
}

int vector_tiles_source_pass(int D, int sdim, int n, bool fact, hipStream_t stream, const KArgs& a, const double* g3, const double* values,
                             const VecTiles& t, const unsigned char* active, double* partial) {
    SourceG g{{g3 ? g3[0] : 0.0, g3 ? g3[1] : 0.0, g3 ? g3[2] : 0.0}};
    int rs = 0;
#define VT_SRC(DV, SV, NV)                                                                                                                              \
    do {                                                                                                                                                \
        if (fact) hipLaunchKernelGGL((k_source_elements_tiled<DV, SV, NV, true, VT_TS>), dim3(8 * ((t.ntiles + 7) / 8)), dim3(VT_TS), 0, stream, a, g, values, t, active, partial); \
        else hipLaunchKernelGGL((k_source_elements_tiled<DV, SV, NV, false, VT_TS>), dim3(8 * ((t.ntiles + 7) / 8)), dim3(VT_TS), 0, stream, a, g, values, t, active, partial);    \
    } while (0)
    if (D == 2 && n == 4) { if (sdim == 1) VT_SRC(2, 1, 4); else VT_SRC(2, 2, 4); }
    else if (D == 2 && n == 3) { if (sdim == 1) VT_SRC(2, 1, 3); else VT_SRC(2, 2, 3); }
    else if (D == 3 && n == 8) { if (sdim == 1) VT_SRC(3, 1, 8); else VT_SRC(3, 3, 8); }
    else if (D == 3 && n == 4) { if (sdim == 1) VT_SRC(3, 1, 4); else VT_SRC(3, 3, 4); }
    else rs = -1;
#undef VT_SRC
    return rs;
}

hipError_t vector_tiles_node_pass(hipStream_t stream, int S, int num_nodes, const VecTiles& t, const double* partial, double* out, const double* scaled_g) {
    const int grid = (num_nodes + 255) / 256;
    if (scaled_g) {   // scalar partials, S components g[c] sum
        const SourceG g{{scaled_g[0], scaled_g[1], scaled_g[2]}};
        if (S == 1) hipLaunchKernelGGL((k_vector_from_partials<1, 1>), dim3(grid), dim3(256), 0, stream, num_nodes, t.np_off, t.np_idx, partial, out, g);
        else if (S == 2) hipLaunchKernelGGL((k_vector_from_partials<1, 2>), dim3(grid), dim3(256), 0, stream, num_nodes, t.np_off, t.np_idx, partial, out, g);
        else hipLaunchKernelGGL((k_vector_from_partials<1, 3>), dim3(grid), dim3(256), 0, stream, num_nodes, t.np_off, t.np_idx, partial, out, g);
        return hipGetLastError();
    }
    if (S == 1) hipLaunchKernelGGL((k_vector_from_partials<1, 0>), dim3(grid), dim3(256), 0, stream, num_nodes, t.np_off, t.np_idx, partial, out);
    else if (S == 2) hipLaunchKernelGGL((k_vector_from_partials<2, 0>), dim3(grid), dim3(256), 0, stream, num_nodes, t.np_off, t.np_idx, partial, out);
    else hipLaunchKernelGGL((k_vector_from_partials<3, 0>), dim3(grid), dim3(256), 0, stream, num_nodes, t.np_off, t.np_idx, partial, out);
    return hipGetLastError();
}

}  // namespace fenris_hip
