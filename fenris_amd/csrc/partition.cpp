// Element partitions of arbitrary meshes behind the C ABI (host side): the index logic a host needs to run one rank of a multi-GPU
// assembly -- node ownership, the extended local mesh with its ring of halo elements, and the lists of interface rows that travel.
// fenris is single-process; this is the multi-process counterpart of handing `CsrParAssembler::assemble_into_csr`
// (src/assembly/global.rs:314-376) one rank's own elements.  fenris_amd/partition.py is the Python mirror (same results, tested
// against each other); fh_group_set_exchange_nodes (group.hip) moves the rows.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <utility>
#include <vector>

#include "../../include/fenris_hip.h"

struct fh_partition {
    std::vector<uint64_t> l2g, elem_l2g, local_conn, owned;
    std::vector<uint8_t> active;
    std::vector<int32_t> send_peers, recv_peers;
    std::vector<uint64_t> send_off, recv_off, send_nodes, recv_nodes;
    uint64_t own_elements = 0, npe = 0;
};

namespace {
inline uint64_t spread21(uint64_t v) {   // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
}  // namespace

extern "C" {

// Default partitioner: elements sorted by the Morton key of their centroids (21 bits per axis), cut into `world` runs of (almost)
// equal length.
static int morton_partition(uint32_t dim, const double* vertices, uint64_t num_vertices, uint64_t npe, const uint64_t* conn, uint64_t E,
                            uint32_t world, int32_t* elem_to_part) {
    if (!vertices || !conn || !elem_to_part || dim < 1 || dim > 3 || npe == 0 || world == 0) return FH_BAD_ARGUMENT;
    std::vector<double> cent((size_t)E * dim);
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (uint64_t e = 0; e < E; ++e)
        for (uint32_t a = 0; a < dim; ++a) {
            double s = 0.0;
            for (uint64_t k = 0; k < npe; ++k) {
                const uint64_t v = conn[e * npe + k];
                if (v >= num_vertices) return FH_BAD_ARGUMENT;
                s += vertices[v * dim + a];
            }
            const double c = s / (double)npe;
            cent[e * dim + a] = c;
            if (e == 0) { lo[a] = c; hi[a] = c; }
            lo[a] = std::min(lo[a], c);
            hi[a] = std::max(hi[a], c);
        }
    std::vector<uint64_t> key((size_t)E, 0);
    for (uint64_t e = 0; e < E; ++e)
        for (uint32_t a = 0; a < dim; ++a) {
            const double scale = hi[a] > lo[a] ? 2097151.0 / (hi[a] - lo[a]) : 0.0;
            key[e] |= spread21((uint64_t)((cent[e * dim + a] - lo[a]) * scale)) << a;
        }
    std::vector<uint64_t> order((size_t)E);
    std::iota(order.begin(), order.end(), 0ull);
    std::stable_sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) { return key[x] < key[y]; });
    for (uint32_t r = 0; r < world; ++r) {
        const uint64_t b0 = (uint64_t)(((unsigned __int128)r * E) / world), b1 = (uint64_t)(((unsigned __int128)(r + 1) * E) / world);
        for (uint64_t i = b0; i < b1; ++i) elem_to_part[order[i]] = (int32_t)r;
    }
    return FH_OK;
}

int fh_morton_partition(uint32_t dim, const double* vertices, uint64_t num_vertices, uint64_t npe, const uint64_t* conn, uint64_t E,
                        uint32_t world, int32_t* elem_to_part) {
    try {
        return morton_partition(dim, vertices, num_vertices, npe, conn, E, world, elem_to_part);
    } catch (...) {
        return FH_HIP_ERROR;   // (out of host memory)
    }
}

// Rank `rank`'s share of a mesh under the element partition elem_to_part (one part in [0, world) per element):
//  * a node is owned by the LOWEST part that has an element touching it (nodes without elements: part 0);
//  * the extended local mesh = the rank's own elements + every element that touches a node they touch; local node ids are the global
//    ids in ascending order (column order is preserved: the rows of every node the rank contributes to carry the global pattern);
//  * active = own elements (halo_mode != 0: also the halo elements that touch an owned node -- the owned rows are then complete
//    without any exchange);
//  * exchange lists (halo_mode == 0): per neighbour the local nodes whose partial rows go there (nodes my elements touch that it
//    owns) and the owned local nodes that receive its partial rows, both in ascending global order.
static fh_partition* partition_create(uint64_t num_nodes, uint64_t npe, const uint64_t* conn, uint64_t E, const int32_t* elem_to_part,
                                      int rank, int world, int halo_mode) {
    if (!conn || !elem_to_part || npe == 0 || world < 1 || rank < 0 || rank >= world) return nullptr;
    constexpr int32_t NONE = std::numeric_limits<int32_t>::max();
    std::vector<int32_t> owner((size_t)num_nodes, NONE);
    for (uint64_t e = 0; e < E; ++e) {
        const int32_t p = elem_to_part[e];
        if (p < 0 || p >= world) return nullptr;
        for (uint64_t k = 0; k < npe; ++k) {
            const uint64_t v = conn[e * npe + k];
            if (v >= num_nodes) return nullptr;
            owner[v] = std::min(owner[v], p);
        }
    }
    for (auto& o : owner) if (o == NONE) o = 0;
    std::vector<uint8_t> touched((size_t)num_nodes, 0);
    uint64_t own = 0;
    for (uint64_t e = 0; e < E; ++e)
        if (elem_to_part[e] == rank) {
            ++own;
            for (uint64_t k = 0; k < npe; ++k) touched[conn[e * npe + k]] = 1;
        }
    std::unique_ptr<fh_partition> P(new fh_partition());
    P->own_elements = own;
    P->npe = npe;
    std::vector<uint8_t> ext_node((size_t)num_nodes, 0);
    for (uint64_t e = 0; e < E; ++e) {
        bool any = false;
        for (uint64_t k = 0; k < npe && !any; ++k) any = touched[conn[e * npe + k]] != 0;
        if (!any) continue;
        P->elem_l2g.push_back(e);
        for (uint64_t k = 0; k < npe; ++k) ext_node[conn[e * npe + k]] = 1;
    }
    std::vector<int64_t> g2l((size_t)num_nodes, -1);
    for (uint64_t v = 0; v < num_nodes; ++v)
        if (ext_node[v] || owner[v] == rank) {   // (isolated owned nodes keep their empty rows)
            g2l[v] = (int64_t)P->l2g.size();
            P->l2g.push_back(v);
        }
    P->local_conn.resize(P->elem_l2g.size() * npe);
    P->active.resize(P->elem_l2g.size());
    for (size_t i = 0; i < P->elem_l2g.size(); ++i) {
        const uint64_t e = P->elem_l2g[i];
        bool act = elem_to_part[e] == rank;
        for (uint64_t k = 0; k < npe; ++k) {
            const uint64_t v = conn[e * npe + k];
            P->local_conn[i * npe + k] = (uint64_t)g2l[v];
            if (halo_mode && owner[v] == rank) act = true;
        }
        P->active[i] = act ? 1 : 0;
    }
    for (uint64_t v = 0; v < num_nodes; ++v)
        if (owner[v] == rank) P->owned.push_back((uint64_t)g2l[v]);
    P->send_off.push_back(0);
    P->recv_off.push_back(0);
    if (!halo_mode) {
        // send: touched nodes owned by q (q < rank necessarily: the owner is the lowest part); ascending global id per peer
        std::vector<std::vector<uint64_t>> snd((size_t)world), rcv((size_t)world);
        for (uint64_t v = 0; v < num_nodes; ++v)
            if (touched[v] && owner[v] != rank) snd[(size_t)owner[v]].push_back((uint64_t)g2l[v]);
        // receive: my nodes that part q's own elements touch -- one pass over the elements, (part, node) pairs sorted and made unique
        std::vector<std::pair<int32_t, uint64_t>> pairs;
        for (uint64_t e = 0; e < E; ++e) {
            const int32_t q = elem_to_part[e];
            if (q == rank) continue;
            for (uint64_t k = 0; k < npe; ++k) {
                const uint64_t v = conn[e * npe + k];
                if (owner[v] == rank) pairs.emplace_back(q, v);
            }
        }
        std::sort(pairs.begin(), pairs.end());
        pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
        for (const auto& qv : pairs) rcv[(size_t)qv.first].push_back((uint64_t)g2l[qv.second]);   // ascending global = ascending local id
        for (int q = 0; q < world; ++q) {
            if (!snd[(size_t)q].empty()) {
                P->send_peers.push_back(q);
                P->send_nodes.insert(P->send_nodes.end(), snd[(size_t)q].begin(), snd[(size_t)q].end());
                P->send_off.push_back(P->send_nodes.size());
            }
            if (!rcv[(size_t)q].empty()) {
                P->recv_peers.push_back(q);
                P->recv_nodes.insert(P->recv_nodes.end(), rcv[(size_t)q].begin(), rcv[(size_t)q].end());
                P->recv_off.push_back(P->recv_nodes.size());
            }
        }
    }
    return P.release();
}

// (no exception crosses the C ABI: out of memory comes back as NULL)
fh_partition* fh_partition_create(uint64_t num_nodes, uint64_t npe, const uint64_t* conn, uint64_t E, const int32_t* elem_to_part,
                                  int rank, int world, int halo_mode) {
    try {
        return partition_create(num_nodes, npe, conn, E, elem_to_part, rank, world, halo_mode);
    } catch (...) {
        return nullptr;
    }
}

void fh_partition_destroy(fh_partition* p) { delete p; }

// sizes: [0] local nodes, [1] local elements, [2] owned nodes, [3] own elements (the unit of the throughput metric),
//        [4] peers to send to, [5] nodes sent in total, [6] peers to receive from, [7] nodes received in total
int fh_partition_sizes(const fh_partition* p, uint64_t sizes[8]) {
    if (!p || !sizes) return FH_BAD_ARGUMENT;
    sizes[0] = p->l2g.size(); sizes[1] = p->elem_l2g.size(); sizes[2] = p->owned.size(); sizes[3] = p->own_elements;
    sizes[4] = p->send_peers.size(); sizes[5] = p->send_nodes.size(); sizes[6] = p->recv_peers.size(); sizes[7] = p->recv_nodes.size();
    return FH_OK;
}

// the extended local mesh: global id of every local node / element, the connectivity in local node ids (-> fh_set_mesh with the
// vertices gathered through l2g), the element mask (-> fh_set_active_elements), the owned local nodes.  NULL skips an array.
int fh_partition_mesh(const fh_partition* p, uint64_t* l2g, uint64_t* elem_l2g, uint64_t* local_conn, uint8_t* active, uint64_t* owned) {
    if (!p) return FH_BAD_ARGUMENT;
    if (l2g) std::copy(p->l2g.begin(), p->l2g.end(), l2g);
    if (elem_l2g) std::copy(p->elem_l2g.begin(), p->elem_l2g.end(), elem_l2g);
    if (local_conn) std::copy(p->local_conn.begin(), p->local_conn.end(), local_conn);
    if (active) std::copy(p->active.begin(), p->active.end(), active);
    if (owned) std::copy(p->owned.begin(), p->owned.end(), owned);
    return FH_OK;
}

// the interface rows: peers (ascending), offsets into the node lists (peers + 1 entries), local node ids per peer -- the arguments of
// fh_group_set_exchange_nodes
int fh_partition_exchange(const fh_partition* p, int32_t* send_peers, uint64_t* send_offsets, uint64_t* send_nodes, int32_t* recv_peers,
                          uint64_t* recv_offsets, uint64_t* recv_nodes) {
    if (!p) return FH_BAD_ARGUMENT;
    if (send_peers) std::copy(p->send_peers.begin(), p->send_peers.end(), send_peers);
    if (send_offsets) std::copy(p->send_off.begin(), p->send_off.end(), send_offsets);
    if (send_nodes) std::copy(p->send_nodes.begin(), p->send_nodes.end(), send_nodes);
    if (recv_peers) std::copy(p->recv_peers.begin(), p->recv_peers.end(), recv_peers);
    if (recv_offsets) std::copy(p->recv_off.begin(), p->recv_off.end(), recv_offsets);
    if (recv_nodes) std::copy(p->recv_nodes.begin(), p->recv_nodes.end(), recv_nodes);
    return FH_OK;
}

}  // extern "C"
