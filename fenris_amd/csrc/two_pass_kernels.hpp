// Second pass of the two-pass owner-computes assembly (engine_two_pass.hip): the row gather from dense element matrices, and the
// per-pattern tables it walks.  Split from assemble_kernels.hpp in round 5 (only engine_two_pass.hip includes it).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

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
// triangle layout: the connectivity with the columns of every element permuted (conn_p[e][perm[n]] = conn[e][n]) and the (node, element) entries with
// the permuted local index
static __global__ void __launch_bounds__(256) k_permute_conn27(long long total, const int* conn, const int* perm, int* conn_p) {
    const long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= total) return;
    const long long e = it / 27;
    conn_p[e * 27 + perm[(int)(it - e * 27)]] = conn[it];
}
static __global__ void __launch_bounds__(256) k_permute_entries27(long long entries, const unsigned* adj, const int* perm, unsigned* adj_p) {
    const long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= entries) return;
    const unsigned ent = adj[it], e = ent / 27u;
    adj_p[it] = e * 27u + (unsigned)perm[ent - e * 27u];
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


// Second pass over element matrices stored as a NODE-BLOCK TRIANGLE (s = 3: blocks of 3 x 3 = 9 contiguous doubles; n (n + 1) / 2 blocks instead of n^2):
//   UPPER (LOWER = false; hex27_blocks.hpp, n = 27):  ke[e][tri(I, J)][i][j], tri(I, J) = I (2 n - 1 - I) / 2 + J for I <= J.  The rows of local node a
//     are the blocks (a, J), J >= a -- one contiguous run -- and, by symmetry (K_e[(a, r), (J, c)] = K_e[(J, c), (a, r)]), the TRANSPOSED blocks (J, a),
//     J < a: a 72-byte piece each.
//   LOWER (the generic first pass, k_assemble_matrix<MODE_DUMP> with ke_tri):  ke[e][J (J + 1) / 2 + I][r][c] = K_e[(J, r), (I, c)], I <= J.  The run is
//     (a, J), J <= a, the transposed pieces are (J, a), J > a.
// A lane takes the entries idx = lane + 64 h of the entry's 9 n values in the order (J, r, c).  Both halves of a symmetric pair of the global matrix
// are sums of the SAME stored doubles in the same (ascending element) order: the assembled matrix is symmetric bit for bit.
template <int n, bool LOWER, typename PT, bool ABL = false>
__global__ void __launch_bounds__(256) k_rows_from_tri(const unsigned* noff, const unsigned* adj_off, const unsigned* adj, const PT* pos_tab,
                                                       const double* ke, double* vals, int overwrite, int max_cnt, const int* node_list,
                                                       int node_count, int npw, int xcw) {
    constexpr int S = 3, TRI = (n * (n + 1) / 2) * 9, EB = 4, HR = (S * S * n + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = (int)(blockDim.x >> 6);
    double* acc = reinterpret_cast<double*>(smem) + (size_t)wave * S * S * max_cnt;
    // the lane's HR places inside an entry (the last round is partly empty; n = 27: 243 = 3 x 64 + 51), as byte offsets into the element's triangle:
    //   direct[h] + 72 row(a)        the block (a, J) of the run      (UPPER: row(a) = a (2 n - 1 - a) / 2, LOWER: a (a + 1) / 2: scalar)
    //   mirror[h] + 72 a             the transposed block (J, a) outside it
    // and the lane of the entry's position load (one load per entry: lane J holds the column slot of local node J; the four places fetch
    // theirs with ds_bpermute) and the place of the value in the staged rows (row r of the node, column c of the block).
    unsigned direct[HR], mirror[HR];
    int Jl[HR], rowoff[HR], colc[HR];
#pragma unroll
    for (int h = 0; h < HR; ++h) {
        const int idx = min(lane + 64 * h, S * S * n - 1);
        const int J = idx / 9, rc = idx - 9 * J;
        Jl[h] = J;
        direct[h] = 8u * (unsigned)idx;
        mirror[h] = 8u * (unsigned)((LOWER ? (J * (J + 1)) / 2 : (J * (2 * n - 1 - J)) / 2) * 9 + (rc % 3) * 3 + rc / 3);
        rowoff[h] = rc / 3;
        colc[h] = rc % 3;
    }
    const bool last_round = lane + 64 * (HR - 1) < S * S * n;
    const unsigned poff = (unsigned)sizeof(PT) * (unsigned)min(lane, n - 1);
    // Everything a node needs before its first value load -- its row range, its range of entries, the entries themselves -- is requested TWO nodes
    // ahead and becomes valid at the one full wait of the node in between (the wait for that node's last values), so that no load is ever
    // waited for across the stores of a finished node (loads and stores retire through one in-order counter).  As dependent loads per node
    // (row offsets -> entry offsets -> entries -> values) the pass spent 1.5 of its 3.5 ms on C4 with every load, add and store taken out.
    //   desc:  lanes 0 .. 3 hold noff[i], noff[i + 1], adj_off[i], adj_off[i + 1]
    //   ents:  lane k holds the node's k-th entry (the first 64; a node with more reads the rest directly)
    // A wavefront takes `npw` CONSECUTIVE nodes (4 on C4: 2 and 4 are level, 8 loses 4 %, 16 and more lose 10 - 20 %, and so do persistent
    // wavefronts over interleaved nodes at any grid: the nodes in flight must stay one compact window of the value array).
    // Workgroups go to the eight XCDs in turn (workgroup b to XCD b mod 8) and every XCD has its own L2: every XCD takes whole chunks of `xcw`
    // consecutive workgroups' worth of nodes (chunk c to XCD c mod 8), so that nodes which share elements -- neighbours in a row, neighbouring
    // rows -- read the lines they share through ONE L2 (C4: 18.1 -> 13.7 GB leave the L2s, -0.2 ms; chunks of 1 024 to 16 384 nodes are level).
    constexpr int step = 1;
    int vb = (int)blockIdx.x;
    if (xcw > 0) {
        const int per = 8 * xcw, full = ((int)gridDim.x / per) * per;
        if (vb < full) { const int xcd = vb & 7, k = vb >> 3; vb = ((k / xcw) * 8 + xcd) * xcw + k % xcw; }
    }
    int it = (vb * wpb + wave) * npw;   // (the four wavefronts on interleaved nodes instead: level)
    node_count = min(node_count, it + npw);
    if (it >= node_count) return;
    auto load_desc = [&](int itx) {
        const int ic = min(itx, node_count - 1);
        const int ii = node_list ? node_list[ic] : ic;
        return ((lane & 2) ? adj_off : noff)[ii + (lane & 1)];
    };
    auto rl = [](unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); };
    auto load_ents = [&](unsigned ta, unsigned tb) { return tb > ta ? adj[ta + min((unsigned)lane, tb - ta - 1u)] : 0u; };
    unsigned d0 = load_desc(it), d1 = load_desc(it + step), d2 = load_desc(it + 2 * step);
    unsigned r0 = rl(d0, 0), r1 = rl(d0, 1), t0 = rl(d0, 2), t1 = rl(d0, 3);
    unsigned n_r0 = rl(d1, 0), n_r1 = rl(d1, 1), n_t0 = rl(d1, 2), n_t1 = rl(d1, 3);
    unsigned ents = load_ents(t0, t1), n_ents = load_ents(n_t0, n_t1);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    for (;;) {
        const int cnt = (int)(r1 - r0);
        if (!(ABL && (overwrite & 0x800))) for (int k = lane; k < S * S * cnt; k += 64) acc[k] = 0.0;
        // requests for the node after the next (its descriptor arrived a node ago) and the descriptor of the one after that
        const unsigned nn_r0 = rl(d2, 0), nn_r1 = rl(d2, 1), nn_t0 = rl(d2, 2), nn_t1 = rl(d2, 3);
        const unsigned nn_ents = load_ents(nn_t0, nn_t1);
        const unsigned d3 = load_desc(it + 3 * step);
        const auto pos_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<PT*>(pos_tab + (size_t)t0 * n), (short)0, (int)((t1 - t0) * (unsigned)(n * sizeof(PT))), 0x00020000);
        for (unsigned t = t0; t < t1; t += EB) {
            // Groups of up to EB entries; every load of the group is issued before the first LDS add.  Only the entries that exist are loaded
            // (corner nodes of a structured mesh have 8 entries, edge nodes 4, face nodes 2, interior nodes 1 -- clamped duplicates were 37 % of
            // the loads of C4), and an entry costs FIVE load instructions: four of values and one of column slots.
            // One straight-line body per group size (a scalar branch picks it): with one `if (k < ng)` around the loads of entry k and another
            // around its adds, the compiler cannot see that the two go together and waits for the loads of entry k - 1 before it touches a
            // register for entry k.
            const int ng = (int)min((unsigned)EB, t1 - t);
            auto group = [&](auto ngc) {
                constexpr int NG = decltype(ngc)::value;
                double v[NG][HR];
                int pl[NG];
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const unsigned tk = t + (unsigned)k;
                    const unsigned ent = tk - t0 < 64u ? rl(ents, (int)(tk - t0)) : (unsigned)__builtin_amdgcn_readfirstlane((int)adj[tk]);
                    const unsigned e = ent / (unsigned)n, a = ent - e * (unsigned)n;
                    const char* kb = reinterpret_cast<const char*>(ke + (size_t)e * TRI);
                    // (a buffer load: scalar base and offset + the lane's constant offset -- as a flat load the compiler builds a 64-bit address in
                    // registers that earlier loads of the group still target, and waits for those loads first)
                    if constexpr (sizeof(PT) == 1) pl[k] = (int)__builtin_amdgcn_raw_buffer_load_b8(pos_rsrc, poff, (int)((tk - t0) * (unsigned)n), 0);
                    else pl[k] = (int)__builtin_amdgcn_raw_buffer_load_b16(pos_rsrc, poff, (int)((tk - t0) * (unsigned)(2 * n)), 0);
                    const unsigned sd = 72u * (LOWER ? (a * (a + 1u)) >> 1 : (a * ((unsigned)(2 * n - 1) - a)) >> 1), sm = 72u * a;
#pragma unroll
                    for (int h = 0; h < HR; ++h) {
                        unsigned off = (LOWER ? (unsigned)Jl[h] <= a : (unsigned)Jl[h] >= a) ? direct[h] + sd : mirror[h] + sm;
                        if (ABL && (overwrite & 0x1000)) off = direct[h] + (unsigned)(72 * n) * (a % (unsigned)((n + 1) / 2));   // (timing only: the entry's 243 values as ONE contiguous run)
                        // (the partly empty last round: loaded under the same condition as it is used -- an unconditional load with a conditional
                        // use is moved down to the use by the compiler, behind the wait for everything else)
                        if (ABL && (overwrite & 0x200)) v[k][h] = 1.0;
                        else if (ABL && (overwrite & 0x2000)) { if (h < HR - 1 || last_round) v[k][h] = __builtin_nontemporal_load(reinterpret_cast<const double*>(kb + off)); }
                        else if (h < HR - 1 || last_round) v[k][h] = *reinterpret_cast<const double*>(kb + off);
                    }
                }
#pragma unroll
                for (int k = 0; k < NG; ++k) {
#pragma unroll
                    for (int h = 0; h < HR; ++h) {
                        const int pos = __builtin_amdgcn_ds_bpermute(4 * Jl[h], pl[k]);
                        if (ABL && (overwrite & 0x400)) { if (pos == 12345 + lane) acc[0] = v[k][h]; }
                        else if (h < HR - 1 || last_round) atomic_add_f64(acc + rowoff[h] * S * cnt + S * pos + colc[h], v[k][h]);
                    }
                }
            };
            if (ng == 4) group(std::integral_constant<int, 4>());
            else if (ng == 2) group(std::integral_constant<int, 2>());
            else if (ng == 1) group(std::integral_constant<int, 1>());
            else group(std::integral_constant<int, 3>());
        }
        // every request of this node has arrived (the last group waited for its values; a node without entries waits here): nothing that the
        // next nodes read is outstanding when the stores go out
        __builtin_amdgcn_s_waitcnt(0x0F70);
        double* out = vals + (size_t)S * S * r0;
        if (ABL && (overwrite & 0x100)) { if (cnt == 12345) out[lane] = acc[lane]; }
        else if (ABL && (overwrite & 0x4000)) for (int k = lane; k < S * S * cnt; k += 64) out[k] = acc[k];
        // (non-temporal: the rows are written once and not read by this kernel -- 0.17 of 3.5 ms on C4)
        else if (overwrite & 1) for (int k = lane; k < S * S * cnt; k += 64) __builtin_nontemporal_store(acc[k], out + k);
        else for (int k = lane; k < S * S * cnt; k += 64) out[k] += acc[k];
        it += step;
        if (it >= node_count) break;
        r0 = n_r0; r1 = n_r1; t0 = n_t0; t1 = n_t1; ents = n_ents;
        n_r0 = nn_r0; n_r1 = nn_r1; n_t0 = nn_t0; n_t1 = nn_t1; n_ents = nn_ents;
        d2 = d3;
    }
}

}  // namespace fenris_hip
