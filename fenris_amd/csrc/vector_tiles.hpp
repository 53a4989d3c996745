// Residual / source vectors through element TILES (vector_tiles.hip, round 4): interface of the translation unit.
// See that file for the design; engine.hip builds the tables once per mesh topology and launches the two passes.
#pragma once
#include <hip/hip_runtime.h>

#include "device_common.hpp"

namespace fenris_hip {


// device tables of one mesh (T tiles, P partial node sums, n nodes per element)
struct VecTiles {
    const int* elem;               // [T][256]        element of every tile thread, ascending inside a tile (-1: none)
    const int* tconn;              // [T][n][256]     node a of the thread's element (threads without one: the tile's first element)
    const unsigned* noff;          // [T + 1]         first partial of every tile
    const unsigned short* la_off;  // [P + T]         tile t: U + 1 starts into its entries, at noff[t] + t
    const unsigned short* la;      // [T][256 n]      entries (thread n + a) of the tile sorted by local node, then ascending
    const unsigned* np_off;        // [N + 1]         node -> its partials ...
    const unsigned* np_idx;        // [P]             ... ascending
    const unsigned* nodes;         // [P]             global node of every partial (a tile's distinct nodes, ascending)
    int ntiles, n;
    unsigned npartials;
    int ts;                        // elements per tile = threads per workgroup of the element pass
};

struct VecTilesStore {
    VecTiles v{};
    void* bufs[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void release();
    ~VecTilesStore() { release(); }
};

// *bad (host) != 0: the tables cannot be built for this mesh (more than 2^32 partials ...): the caller keeps the two-pass kernels
hipError_t vector_tiles_build(hipStream_t stream, const int* conn, int n, long long E, const double* verts, int D, int num_nodes, VecTilesStore* out,
                              int* bad);

// element pass: every tile's elements in registers, their vectors summed per distinct node of the tile into partial[P][S].
// active (device, may be null): elements with active[e] == 0 contribute nothing.  Returns -1 when (elem_kind, op) is not covered.
int vector_tiles_element_pass(int elem_kind, int op, hipStream_t stream, const KArgs& a, const VecTiles& t, const unsigned char* active, double* partial);

// energy over the tiles: one partial per workgroup into partial[] (the number of workgroups = partials comes back; -1: not covered);
// the caller sums them in index order (k_sum_partials)
int vector_tiles_energy_pass(int elem_kind, int op, hipStream_t stream, const KArgs& a, const VecTiles& t, const unsigned char* active, double* partial);
inline int vector_tiles_energy_partials(const VecTiles& t) { return 8 * ((t.ntiles + 7) / 8); }

// source vector (k_source_elements' arithmetic) over the tiles; g3: three doubles (host) or null; fact: scalar partials (GravitySource),
// the node pass multiplies by g.  Returns -1 when (D, n) is not covered.
int vector_tiles_source_pass(int D, int sdim, int n, bool fact, hipStream_t stream, const KArgs& a, const double* g3, const double* values,
                             const VecTiles& t, const unsigned char* active, double* partial);

// node pass: out[S node + c] += sum of the node's partials in ascending order (scaled_g != null: scalar partials, out += g[c] sum)
hipError_t vector_tiles_node_pass(hipStream_t stream, int S, int num_nodes, const VecTiles& t, const double* partial, double* out,
                                  const double* scaled_g = nullptr);

}  // namespace fenris_hip
