// Element colouring ON the device (the reference colours on the host: color_nodes, src/assembly/global.rs:540-551, with
// sequential_greedy_coloring, fenris-paradis/src/coloring.rs:6-70 -- fh_color restates that one exactly).  This one is the parallel
// counterpart for meshes that live on the device: rounds of "propose the smallest colour no finished neighbour holds" / "keep it unless a
// neighbour with a smaller (hashed) priority proposed the same" -- Luby-style, deterministic (no round reads what the same round writes),
// O(log E) rounds in expectation.  Two elements are neighbours when they share a node; a colour is a set of pairwise non-neighbours, which
// is all DisjointSubsets / CsrParAssembler need (fenris-paradis/src/lib.rs, global.rs:314-376).  It is generally NOT the sequential
// greedy colouring: the colour count may be larger (a structured Hex8 mesh: 8 sequentially, 18 here).
#pragma once
#include <hip/hip_runtime.h>

namespace fenris_hip {

constexpr int COLOR_WINDOW = 128;      // colours one pass over the neighbours can mark
constexpr int COLOR_LIMIT = 1 << 15;   // more would need more windows than anyone waits for (a node with > 32 k elements)

__device__ __forceinline__ unsigned color_priority(unsigned e) {   // murmur3 finaliser: a fixed pseudo-random order of the elements
    e ^= e >> 16; e *= 0x85ebca6bu; e ^= e >> 13; e *= 0xc2b2ae35u; e ^= e >> 16;
    return e;
}

// n2e: node -> entries v = element * n + local node, sorted (k_sort_n2e)
static __global__ void __launch_bounds__(256) k_color_propose(int E, int n, const int* conn, const unsigned* n2e_off, const unsigned* n2e,
                                                      const int* color, int* tent, int* overflow) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    if (color[e] >= 0) { tent[e] = -1; return; }
    int pick = -1;
    // windows of 128 colours: almost always the first one holds a free colour; a node with hundreds of elements walks on
    for (int wb = 0; wb < COLOR_LIMIT && pick < 0; wb += COLOR_WINDOW) {
        unsigned long long forb[2] = {0ull, 0ull};
        for (int a = 0; a < n; ++a) {
            const unsigned node = (unsigned)conn[(size_t)e * n + a];
            for (unsigned k = n2e_off[node]; k < n2e_off[node + 1]; ++k) {
                const int cf = color[n2e[k] / (unsigned)n] - wb;
                if (cf >= 0 && cf < COLOR_WINDOW) forb[cf >> 6] |= 1ull << (cf & 63);
            }
        }
        if (~forb[0]) pick = wb + __ffsll((long long)~forb[0]) - 1;
        else if (~forb[1]) pick = wb + 64 + __ffsll((long long)~forb[1]) - 1;
    }
    if (pick < 0) atomicOr(overflow, 1);
    tent[e] = pick;
}

static __global__ void __launch_bounds__(256) k_color_resolve(int E, int n, const int* conn, const unsigned* n2e_off, const unsigned* n2e,
                                                      const int* tent, int* color, unsigned* remaining) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int t = tent[e];
    if (t < 0) return;
    const unsigned pe = color_priority((unsigned)e);
    bool keep = true;
    for (int a = 0; a < n && keep; ++a) {
        const unsigned node = (unsigned)conn[(size_t)e * n + a];
        for (unsigned k = n2e_off[node]; k < n2e_off[node + 1]; ++k) {
            const unsigned f = n2e[k] / (unsigned)n;
            if ((int)f == e || tent[f] != t) continue;
            const unsigned pf = color_priority(f);
            if (pf < pe || (pf == pe && (int)f < e)) { keep = false; break; }
        }
    }
    if (keep) { color[e] = t; atomicMax(remaining + 1, (unsigned)t); }
    else atomicAdd(remaining, 1u);
}

static __global__ void __launch_bounds__(256) k_color_iota(int E, unsigned* ids, int* color, int fill) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    ids[e] = (unsigned)e;
    color[e] = fill;
}

}  // namespace fenris_hip
