// Owner-computes partition: node blocks, position tables, lane tables (once per pattern)
#include "engine_internal.hpp"

// Lane tables of the row-owner kernels (k_affine_rows, k_hex8_rows) for the positions described by the pipelined kernel's records
// `rec`: one record of 256 lanes per position (affine_rows_build), positions with identical records share one table (hashed on the
// device with two independent 64-bit hashes, merged here on equality of the 128 bits; only the full form (FENRIS_HIP_NO_LANE_DEDUPE, or more than 2^23 positions) still
// compares the records on the device), the table id goes into every header.  `bad`: some block cannot be expressed.
static int build_lane_tables(fh_ctx* c, const int* rec, int us, int ms, int nb_target, int npos, int S, const int* conn, const int* elem,
                             DevBuf<int4>& hdr, DevBuf<uint2>& lanes, int& ntab_out, int& incomplete_out, bool& bad_out, const char* what,
                             int mirror = 0) {
    DevBuf<int> st;
    HIP_TRY(c, st.alloc(2));
    HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
    HIP_TRY(c, hdr.alloc((size_t)npos));
    // Hash-only build (round 4): the 256 records of a position are formed in LDS, hashed twice (128 bits) and dropped; the records of the
    // first position of every distinct table are formed once more into the compact tables.  Writing all of them (2 KB x 1.46 M positions
    // = 3 GB on the 216^3 mesh) cost an allocation of 40 - 120 ms.  FENRIS_HIP_NO_LANE_DEDUPE keeps the full form (its compaction
    // compares every position with its table); two positions whose first hashes agree and whose second ones differ send the build there too.
    bool full = c->env("FENRIS_HIP_NO_LANE_DEDUPE") != nullptr || npos >= (1 << 23);
    DevBuf<uint2> lanes_full;
    DevBuf<unsigned long long> hash_d;
    HIP_TRY(c, hash_d.alloc((size_t)npos * 2));
    HostBuf<unsigned long long> hash_h((size_t)npos * 2);   // (host_pool.hpp: not a std::vector)
    int bad = 0;
    if (!full) {
        HIP_TRY(c, affine_rows_build(c->stream, rec, c->p_rw, us, ms, nb_target, npos, S, c->ncols.p, conn, c->p_cs, elem, hdr.p, (uint2*)nullptr, st.p,
                                     hash_d.p, mirror, hash_d.p + npos, (int)c->max_row));
        HIP_TRY(c, hipMemcpyAsync(&bad, st.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(hash_h.data(), hash_d.p, sizeof(unsigned long long) * (size_t)npos * 2, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        bad_out = bad != 0;
        if (bad) return FH_OK;
        HostBuf<int> ids((size_t)npos);
        hvec<int> first;
        std::unordered_map<unsigned long long, int> seen;
        seen.reserve(1024);
        bool collision = false;
        for (int p = 0; p < npos && !collision; ++p) {
            auto it = seen.find(hash_h[p]);
            if (it == seen.end()) {
                it = seen.emplace(hash_h[p], (int)first.size()).first;
                first.push_back(p);
            } else if (hash_h[(size_t)npos + first[it->second]] != hash_h[(size_t)npos + p]) {
                collision = true;
            }
            ids[p] = it->second;
        }
        if (!collision) {
            const int ntab = (int)first.size();
            DevBuf<int> ids_d, first_d;
            HIP_TRY(c, ids_d.alloc((size_t)npos));
            HIP_TRY(c, first_d.alloc((size_t)ntab));
            HIP_TRY(c, lanes.alloc((size_t)ntab * 256));
            HIP_TRY(c, hipMemcpyAsync(ids_d.p, ids.data(), sizeof(int) * (size_t)npos, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(first_d.p, first.data(), sizeof(int) * (size_t)ntab, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
            HIP_TRY(c, affine_rows_tables(c->stream, rec, c->p_rw, us, ms, nb_target, npos, S, c->ncols.p, conn, c->p_cs, elem, mirror, ids_d.p, first_d.p,
                                          ntab, lanes.p, hdr.p, st.p, (int)c->max_row));
            int mismatch[2] = {0, 0};   // [1]: some position has a block without an owner lane
            HIP_TRY(c, hipMemcpyAsync(mismatch, st.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            ntab_out = ntab;
            incomplete_out = mismatch[1];
            if (c->env("FENRIS_HIP_VERBOSE")) {
                long long changes = 0;
                for (int p = 1; p < npos; ++p) changes += ids[p] != ids[p - 1];
                std::fprintf(stderr, "[fenris_hip] %s: %d positions share %d lane tables, %lld changes of table along the sweep%s\n", what,
                             npos, ntab_out, changes, incomplete_out ? ", some position has a block without an owner" : "");
            }
            return FH_OK;
        }
        full = true;
        HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
    }
    HIP_TRY(c, lanes_full.alloc((size_t)npos * 256));
    HIP_TRY(c, affine_rows_build(c->stream, rec, c->p_rw, us, ms, nb_target, npos, S, c->ncols.p, conn, c->p_cs, elem, hdr.p, lanes_full.p, st.p,
                                 hash_d.p, mirror, nullptr, (int)c->max_row));
    HIP_TRY(c, hipMemcpyAsync(&bad, st.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hash_h.data(), hash_d.p, sizeof(unsigned long long) * (size_t)npos, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    bad_out = bad != 0;
    if (bad) return FH_OK;
    // positions with identical lane records (the interior of a structured mesh) share one table: the kernel
    // skips the fetch when the table does not change, and what it fetches stays in the caches
    HostBuf<int> ids((size_t)npos);
    hvec<int> first;
    auto dedupe = [&](bool identity) {
        first.clear();
        if (identity) {
            first.resize((size_t)npos);
            for (int p = 0; p < npos; ++p) { ids[p] = p; first[p] = p; }
            return;
        }
        std::unordered_map<unsigned long long, int> seen;
        seen.reserve(1024);
        for (int p = 0; p < npos; ++p) {
            auto it = seen.find(hash_h[p]);
            if (it == seen.end()) {
                it = seen.emplace(hash_h[p], (int)first.size()).first;
                first.push_back(p);
            }
            ids[p] = it->second;
        }
    };
    dedupe(c->env("FENRIS_HIP_NO_LANE_DEDUPE") != nullptr || npos >= (1 << 23));  // the id has 23 bits
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int ntab = (int)first.size();
        DevBuf<int> ids_d, first_d;
        HIP_TRY(c, ids_d.alloc((size_t)npos));
        HIP_TRY(c, first_d.alloc((size_t)ntab));
        HIP_TRY(c, lanes.alloc((size_t)ntab * 256));
        HIP_TRY(c, hipMemcpyAsync(ids_d.p, ids.data(), sizeof(int) * (size_t)npos, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(first_d.p, first.data(), sizeof(int) * (size_t)ntab, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
        HIP_TRY(c, affine_rows_compact(c->stream, lanes_full.p, ids_d.p, first_d.p, npos, ntab, lanes.p, hdr.p, st.p));
        int mismatch[2] = {0, 0};   // [1]: some position has a block without an owner lane
        HIP_TRY(c, hipMemcpyAsync(mismatch, st.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        ntab_out = ntab;
        incomplete_out = mismatch[1];
        if (!mismatch[0]) break;
        dedupe(true);  // a hash collision: every position keeps its own table
    }
    if (c->env("FENRIS_HIP_VERBOSE")) {
        long long changes = 0;   // positions whose table differs from their predecessor's in the sweep: each is a 2 KB fetch
        for (int p = 1; p < npos; ++p) changes += ids[p] != ids[p - 1];
        std::fprintf(stderr, "[fenris_hip] %s: %d positions share %d lane tables, %lld changes of table along the sweep%s\n", what,
                     npos, ntab_out, changes, incomplete_out ? ", some position has a block without an owner" : "");
    }
    return FH_OK;
}

// largest number of unique elements / of (node, element) entries over the block headers
static __global__ void __launch_bounds__(256) k_gather_hdr_max(const GatherHdr* hdr, int nblk, int* out) {
    int u = 1, m = 1, r = 1;   // ... and of node-level entries of the block's rows
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += gridDim.x * blockDim.x) { u = max(u, hdr[b].U); m = max(m, hdr[b].m); r = max(r, hdr[b].nrow); }
    for (int o = 32; o > 0; o >>= 1) { u = max(u, __shfl_xor(u, o)); m = max(m, __shfl_xor(m, o)); r = max(r, __shfl_xor(r, o)); }
    if ((threadIdx.x & 63) == 0) { atomicMax(out, u); atomicMax(out + 1, m); atomicMax(out + 2, r); }
}

// The cut of the node range into owner blocks for numberings made of LONG runs (grid lines), on the device (round 5): one thread per run start
// walks its run and cuts it into ceil(L / nb_target) balanced pieces -- exactly what the host loop of build_partition does for such a run.
// Anything else (a run shorter than nb_target, a piece over the budgets, a run too long for one thread to walk) raises info[0] and the host
// loop does the whole cut.  start[n] = 1 for the first node of every piece and for n_hi; info[1] / info[2]: largest number of node-level
// entries / of (node, element) adjacencies of a piece.
static __global__ void __launch_bounds__(256) k_cut_runs(const unsigned char* link, const unsigned* noff, const unsigned* adj_off, int n_lo, int n_hi,
                                                         int nb_target, long long acc_entries, int mb, unsigned char* start, int* info) {
    const int i = n_lo + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_hi) return;
    if (i != n_lo && link[i - 1]) return;   // inside a run
    int r1 = i + 1;
    while (r1 < n_hi && link[r1 - 1] && r1 - i <= 65536) ++r1;
    const int L = r1 - i;
    if (L < nb_target || L > 65536) { info[0] = 1; return; }
    const int k = (L + nb_target - 1) / nb_target;
    int prev = i, mx = 0, mm = 0;
    for (int j = 1; j <= k; ++j) {
        const int e = i + (int)((long long)L * j / k);
        const long long rows = (long long)noff[e] - (long long)noff[prev], ent = (long long)adj_off[e] - (long long)adj_off[prev];
        if (rows > acc_entries || ent > mb) { info[0] = 1; return; }
        mx = max(mx, (int)rows);
        mm = max(mm, (int)ent);
        start[prev] = 1;
        prev = e;
    }
    if (r1 == n_hi) start[n_hi] = 1;
    atomicMax(info + 1, mx);
    atomicMax(info + 2, mm);
}

static __global__ void __launch_bounds__(256) k_iota_int(int* out, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = i;
}

// the lane tuner of k_hex8_rows on the context's unique lane tables (hex8_rows.hip: simulated annealing over moves that leave every sum unchanged);
// called by build_partition or, deferred, in front of a later launch (fh_ctx::h_tune_pending)
int hex8_tune_lanes_now(fh_ctx* c) {
    c->h_tune_pending = 0;
    if (!c->h_lanes.p || c->h_ntab <= 0) return FH_OK;
    const auto t0 = std::chrono::steady_clock::now();
    HostBuf<uint2> tabs((size_t)c->h_ntab * 256);
    HIP_TRY(c, hipMemcpyAsync(tabs.data(), c->h_lanes.p, sizeof(uint2) * tabs.size(), hipMemcpyDeviceToHost, c->stream));
    // The tables are rewritten in place: NO launch that reads them may still be running -- on any stream (fh_set_stream does not drain the
    // stream it replaces, so a k_hex8_rows of this context can be in flight on another one).  Once per pattern, next to ~18 ms of host work.
    HIP_TRY(c, hipDeviceSynchronize());
    double cb = 0.0, ca = 0.0;
    hex8_rows_tune_lanes(tabs.data(), c->h_ntab, 12345u, &cb, &ca, 1000000ll);   // (proposals: a third of round 4's, +1.7 % kernel time for half the tuner's)
    HIP_TRY(c, hipMemcpyAsync(c->h_lanes.p, tabs.data(), sizeof(uint2) * tabs.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->env("FENRIS_HIP_VERBOSE"))
        std::fprintf(stderr, "[fenris_hip] hex8 rows: %d lane tables tuned in %.1f ms, modelled LDS cycles per position and operand sweep %.1f -> %.1f (64 = conflict-free)\n",
                     c->h_ntab, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), cb, ca);
    return FH_OK;
}

// greedy partition of the node range into owner blocks (gather mode)

int build_partition(fh_ctx* c) {
    if (c->has_partition) return FH_OK;
    // FENRIS_HIP_VERBOSE: wall time of the stages of this set-up (stream drained at every mark)
    auto t_last = std::chrono::steady_clock::now();
    const bool vt = c->env("FENRIS_HIP_VERBOSE") != nullptr;
    // (The stream is drained at every stage boundary ALSO without the print: measured on the 216^3 mesh, the first assembly takes 90 ms
    // with these synchronisations and 118 ms without them -- the stages' temporaries are released with work still queued behind them
    // otherwise, and a release then waits out the whole queue inside the runtime.)
    const bool stage_sync = true;
    auto mark = [&](const char* what) {
        if (stage_sync || vt) (void)hipStreamSynchronize(c->stream);
        if (!vt) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[fenris_hip] set-up: %-34s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // (round 5: the host copies of the two offset arrays -- 82 MB over PCIe on the 216^3 mesh -- are made only where the host cuts the node
    // range itself; numberings made of grid lines are cut on the device, k_cut_runs)
    // adjacency that drives the numerics: all elements, or only the active ones when a mask is set
    const std::vector<unsigned>& adj_off_h = c->has_mask ? c->h_n2e_off_c : c->h_n2e_off;
    const unsigned* adj_off_d = c->has_mask ? c->n2e_off_c.p : c->n2e_off.p;
    const unsigned* adj_d = c->has_mask ? c->n2e_c.p : c->n2e.p;
    const int S = c->S();
    const int N = (int)c->N;
    const unsigned max_row = c->max_row;
    // The owner blocks are ranges of consecutive nodes.  On a mesh whose numbering has no locality (consecutive nodes share no
    // element: every node of a block brings its own ~24 tetrahedra, C3: 128 slots and 11 kB of vertex gathers for 5 nodes) the
    // row-owner Tet4 kernel -- whose lanes store every block of a row straight to its place, so that the rows of a block
    // need not be neighbours in memory -- gets its blocks from a locality order instead: nodes sorted by the Morton key of
    // their coordinates, the pattern rows and the node -> element adjacency permuted alike (contents unchanged: real node
    // and element ids), the real first entry of every row handed to the kernel (r_rec).  Everything below then works on
    // positions in that order; nothing else in the context sees it.
    const unsigned* noff_d = c->noff.p;
    const unsigned* ncols_d = c->ncols.p;
    const std::vector<unsigned>* h_noff_p = &c->h_noff;
    const std::vector<unsigned>* adj_off_hp = &adj_off_h;
    DevBuf<unsigned> v2r_d, noff_v, ncols_v, adj_off_v, adj_v;
    DevBuf<int> r2v_d;
    std::vector<unsigned> h_noff_v, adj_off_hv;
    c->part_perm = false;
    const bool perm_cand = c->elem_kind == FH_TET4 && (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC) && c->row_hi < 0 &&
                           !c->perm_failed && !c->has_rules && c->fast_ok && N > 64 && !c->env("FENRIS_HIP_NO_ROWS") &&
                           !c->env("FENRIS_HIP_NO_NODE_ORDER") && !c->env("FENRIS_HIP_TRACE");
    if (perm_cand) {
        // how local is the numbering?  fraction of nodes that share an element with their successor
        DevBuf<unsigned char> link_d;
        HIP_TRY(c, link_d.alloc((size_t)N + 1));
        hipLaunchKernelGGL(k_linked_to_next, dim3((N + 255) / 256), dim3(256), 0, c->stream, adj_off_d, adj_d, c->ei.n, N, link_d.p);
        HostBuf<unsigned char> lk((size_t)N);
        HIP_TRY(c, hipMemcpyAsync(lk.data(), link_d.p, (size_t)N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        long long linked = 0;
        for (int i = 0; i < N; ++i) linked += lk[i];
        const bool force = c->env_int("FENRIS_HIP_NODE_ORDER", 0) != 0;
        if (force || linked * 2 < (long long)N) {
            { const int rc_h = host_offsets(c); if (rc_h) return rc_h; }
            const int D = c->ei.d;
            HostBuf<double> hv((size_t)N * D);
            HIP_TRY(c, hipMemcpyAsync(hv.data(), c->verts.p, sizeof(double) * hv.size(), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
            for (int k = 0; k < D; ++k) { lo[k] = hv[k]; hi[k] = hv[k]; }
            for (size_t i = 0; i < (size_t)N; ++i)
                for (int k = 0; k < D; ++k) { lo[k] = std::min(lo[k], hv[i * D + k]); hi[k] = std::max(hi[k], hv[i * D + k]); }
            for (int k = 0; k < D; ++k) sc[k] = (hi[k] > lo[k]) ? 2097151.0 / (hi[k] - lo[k]) : 0.0;
            DevBuf<unsigned long long> keys, keys_s;
            DevBuf<unsigned> ids;
            HIP_TRY(c, keys.alloc((size_t)N));
            HIP_TRY(c, keys_s.alloc((size_t)N));
            HIP_TRY(c, ids.alloc((size_t)N));
            HIP_TRY(c, v2r_d.alloc((size_t)N));
            hipLaunchKernelGGL(k_morton_keys, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->verts.p, N, D, lo[0], lo[1], lo[2], sc[0],
                               sc[1], sc[2], keys.p, ids.p);
            size_t tb = 0;
            HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys.p, keys_s.p, ids.p, v2r_d.p, N, 0, 64, c->stream));
            DevBuf<char> tmp;
            HIP_TRY(c, tmp.alloc(tb + 16));
            HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, keys.p, keys_s.p, ids.p, v2r_d.p, N, 0, 64, c->stream));
            HIP_TRY(c, r2v_d.alloc((size_t)N));
            hipLaunchKernelGGL(k_invert_perm, dim3((N + 255) / 256), dim3(256), 0, c->stream, v2r_d.p, N, r2v_d.p);
            // rows of the pattern and of the adjacency in that order
            auto permute_rows = [&](const unsigned* off_src, const unsigned* src, size_t total, DevBuf<unsigned>& off_dst, DevBuf<unsigned>& dst,
                                    std::vector<unsigned>& off_h) -> int {
                DevBuf<unsigned> len;
                HIP_TRY(c, len.alloc((size_t)N + 1));
                HIP_TRY(c, off_dst.alloc((size_t)N + 1));
                HIP_TRY(c, dst.alloc(total + 1));
                hipLaunchKernelGGL(k_perm_row_lengths, dim3((N + 256) / 256), dim3(256), 0, c->stream, off_src, v2r_d.p, N, len.p);
                size_t sb = 0;
                HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, sb, len.p, off_dst.p, N + 1, c->stream));
                DevBuf<char> t2;
                HIP_TRY(c, t2.alloc(sb + 16));
                HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(t2.p, sb, len.p, off_dst.p, N + 1, c->stream));
                hipLaunchKernelGGL(k_perm_copy_rows, dim3((N + 255) / 256), dim3(256), 0, c->stream, off_src, src, v2r_d.p, off_dst.p, dst.p, N);
                HIP_TRY(c, hipGetLastError());
                off_h.resize((size_t)N + 1);
                HIP_TRY(c, hipMemcpyAsync(off_h.data(), off_dst.p, sizeof(unsigned) * ((size_t)N + 1), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));  // len / t2 are released on return
                return FH_OK;
            };
            int rp = permute_rows(c->noff.p, c->ncols.p, (size_t)c->h_noff[N], noff_v, ncols_v, h_noff_v);
            if (rp) return rp;
            rp = permute_rows(adj_off_d, adj_d, (size_t)adj_off_h[N], adj_off_v, adj_v, adj_off_hv);
            if (rp) return rp;
            noff_d = noff_v.p;
            ncols_d = ncols_v.p;
            adj_off_d = adj_off_v.p;
            adj_d = adj_v.p;
            h_noff_p = &h_noff_v;
            adj_off_hp = &adj_off_hv;
            c->part_perm = true;
            if (c->env("FENRIS_HIP_VERBOSE"))
                std::fprintf(stderr, "[fenris_hip] node numbering without locality (%.1f %% of the nodes share an element with their successor): "
                                     "owner blocks formed in Morton order\n", 100.0 * (double)linked / (double)N);
        }
    }
    const std::vector<unsigned>& h_noff = *h_noff_p;
    const std::vector<unsigned>& adj_off_hh = *adj_off_hp;
    // nodes per block (tunable), entry capacity per batch, accumulator budget
    // Hex8 meshes with affine elements: 36 row lanes per node in k_affine_rows, seven nodes per block also for S = 1
    const bool aff_cand = c->elem_kind == FH_HEX8 && c->has_aff && c->num_aff > 0 && !c->aff_failed && c->affine_tol > 0.0 &&
                          (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC || (c->op == FH_MASS_SCALAR && c->has_params));
    const bool rows_special = perm_cand;   // tables for the row-owner Tet4 kernel alone: larger blocks (below)
    // Hex8 Laplace / LinearElastic without a mask: the general positions run on k_hex8_rows (36 row lanes per node as well)
    const bool hrows_cand = c->elem_kind == FH_HEX8 && (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC) && !c->has_rules &&
                            !c->env("FENRIS_HIP_NO_HEX8_ROWS");
    // (row-owner Tet4 tables, by attempt: 13 nodes / 352 entries, 11 / 288, 9 / 256, 7 / 224 -- whatever the lane format takes: 256 lanes, 252 slots,
    // 256 distinct vertices per position; round 6: a position's cost is mostly cost per POSITION, C3 0.456 ms at 9 nodes, 0.394 at 12 - 13)
    static const int rows_nb[4] = {13, 11, 9, 7}, rows_mb[4] = {352, 288, 256, 224};
    const int nb_target = std::max(1, std::min(64, c->env_int("FENRIS_HIP_GATHER_NB", rows_special ? rows_nb[std::min(c->rows_try, 3)]
                                                                                                     : (S == 1 && !aff_cand && !hrows_cand) ? 8 : 7)));  // < 256: packed in 8 bits
    // Tables for the row-owner Tet4 kernel alone may hold more entries per block than the pipelined kernel's lane mapping takes
    // and more nodes (the lane word has four bits for the node): nine nodes / 256 entries first (C3: 98 k positions of ~170 lanes
    // instead of 171 k of ~90, 0.80 -> 0.64 ms), seven / 224 when that cannot be expressed (0.67 ms), then the standard form
    c->part_rows_only = rows_special;
    const int mb = std::max(16, std::min(1024, c->env_int("FENRIS_HIP_GATHER_MB", rows_special ? rows_mb[std::min(c->rows_try, 3)] : 128)));
    const size_t lds_target = (size_t)c->env_int("FENRIS_HIP_GATHER_LDS_KB", 52) * 1024;
    // accumulators: nb_target typical rows, but at least the largest single row block
    unsigned noff_ends[2] = {0, 0};   // (telescoping sum of the row lengths; two values of the device array: the host copy may not exist)
    if (N) {
        HIP_TRY(c, hipMemcpyAsync(&noff_ends[0], noff_d, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&noff_ends[1], noff_d + N, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    const long long sum_rows = (long long)noff_ends[1] - (long long)noff_ends[0];
    const int avg_row = N ? (int)((sum_rows + N - 1) / N) : 1;
    int acc = S * S * std::max<int>((int)max_row, std::min<int>(nb_target * (avg_row + avg_row / 4 + 1), 8192 / (S * S)));
    hvec<unsigned> blk;
    // owner-computes covers the nodes [n_lo, n_hi): everything, or the range of fh_set_row_range
    const int n_lo = (c->row_hi < 0) ? 0 : (int)std::min<long long>(c->row_lo, N);
    const int n_hi = (c->row_hi < 0) ? N : (int)std::min<long long>(c->row_hi, N);
    blk.push_back((unsigned)n_lo);
    // Blocks are aligned to RUNS of consecutive nodes that share an element with their successor (the grid lines of
    // a structured numbering): a run of L >= nb_target nodes is cut into ceil(L / nb_target) blocks of balanced size,
    // so that every line of a structured mesh is cut at the same places and consecutive blocks of a sweep chain
    // share exactly the elements between two lines.  Short runs (unstructured numberings) are merged greedily.
    hvec<unsigned char> link;
    DevBuf<unsigned char> link_d;
    const bool aligned = N > 0;
    if (aligned) {
        HIP_TRY(c, link_d.alloc((size_t)N + 1));
        hipLaunchKernelGGL(k_linked_to_next, dim3((N + 255) / 256), dim3(256), 0, c->stream, adj_off_d, adj_d, c->ei.n, N, link_d.p);
        HIP_TRY(c, hipGetLastError());
    }
    // Numberings made of grid lines: the cut on the device (k_cut_runs), the block offsets never on the host
    bool cut_done = false;
    unsigned max_m = 0;
    if (aligned && n_hi > n_lo && c->env_int("FENRIS_HIP_HOST_CUT", 0) == 0) {
        DevBuf<unsigned char> start_d;
        DevBuf<int> info_d;
        HIP_TRY(c, start_d.alloc((size_t)N + 2));
        HIP_TRY(c, info_d.alloc(4));
        HIP_TRY(c, hipMemsetAsync(start_d.p, 0, (size_t)N + 2, c->stream));
        HIP_TRY(c, hipMemsetAsync(info_d.p, 0, sizeof(int) * 4, c->stream));
        hipLaunchKernelGGL(k_cut_runs, dim3((unsigned)((n_hi - n_lo + 255) / 256)), dim3(256), 0, c->stream, link_d.p, noff_d, adj_off_d, n_lo, n_hi, nb_target,
                           (long long)(acc / (S * S)), mb, start_d.p, info_d.p);
        HIP_TRY(c, hipGetLastError());
        int info[4] = {1, 0, 0, 0};
        HIP_TRY(c, hipMemcpyAsync(info, info_d.p, sizeof info, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!info[0]) {
            const int span = n_hi - n_lo + 1;
            DevBuf<unsigned> sel;
            HIP_TRY(c, sel.alloc((size_t)span + 1));
            hipcub::CountingInputIterator<unsigned> first((unsigned)n_lo);
            size_t tb = 0;
            HIP_TRY(c, hipcub::DeviceSelect::Flagged(nullptr, tb, first, start_d.p + n_lo, sel.p, info_d.p + 3, span, c->stream));
            DevBuf<char> tmp;
            HIP_TRY(c, tmp.alloc(tb + 16));
            HIP_TRY(c, hipcub::DeviceSelect::Flagged(tmp.p, tb, first, start_d.p + n_lo, sel.p, info_d.p + 3, span, c->stream));
            int count = 0;
            HIP_TRY(c, hipMemcpyAsync(&count, info_d.p + 3, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (count >= 2) {
                HIP_TRY(c, c->blk_off.alloc((size_t)count));
                HIP_TRY(c, hipMemcpyAsync(c->blk_off.p, sel.p, sizeof(unsigned) * (size_t)count, hipMemcpyDeviceToDevice, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));   // sel / tmp are released at the end of this scope
                c->nblk = count - 1;
                acc = S * S * std::max(1, info[1]);   // the accumulator budget tightened to the largest block actually formed
                max_m = (unsigned)info[2];
                cut_done = true;
            }
        }
    }
    if (!cut_done) {
    { const int rc_h = host_offsets(c); if (rc_h) return rc_h; }
    link.assign((size_t)N + 1, 1);
    if (aligned) {
        HIP_TRY(c, hipMemcpyAsync(link.data(), link_d.p, (size_t)N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    mark("row sums, link flags");
    auto fits = [&](int a, int b) {  // nodes [a, b) within the accumulator and entry budgets
        return S * S * ((long long)h_noff[b] - h_noff[a]) <= acc && (long long)adj_off_hh[b] - adj_off_hh[a] <= mb;
    };
    auto cut_greedy = [&](int a, int b) {  // blocks of up to nb_target nodes, shrunk where the budgets demand it
        while (a < b) {
            int e = std::min(b, a + nb_target);
            while (e > a + 1 && !fits(a, e)) --e;
            blk.push_back((unsigned)e);
            a = e;
        }
    };
    int i0 = n_lo;
    while (i0 < n_hi) {
        int r1 = i0 + 1;  // end of the run that starts at i0
        while (r1 < n_hi && link[r1 - 1]) ++r1;
        const int L = r1 - i0;
        if (L >= nb_target) {
            const int k = (L + nb_target - 1) / nb_target;
            bool ok = true;
            std::vector<int> ends;
            for (int j = 1; j <= k && ok; ++j) {
                const int e = i0 + (int)((long long)L * j / k);
                ok = fits(ends.empty() ? i0 : ends.back(), e);
                ends.push_back(e);
            }
            if (ok) for (int e : ends) blk.push_back((unsigned)e);
            else cut_greedy(i0, r1);
            i0 = r1;
        } else {
            // short runs (unstructured numbering): the whole stretch up to the next long run is cut greedily
            int e = r1;
            while (e < n_hi) {
                int r2 = e + 1;
                while (r2 < n_hi && link[r2 - 1]) ++r2;
                if (r2 - e >= nb_target) break;
                e = r2;
            }
            cut_greedy(i0, e);
            i0 = e;
        }
    }
    mark("cutting the node range (host)");
    c->nblk = (int)blk.size() - 1;
    }   // !cut_done
    else mark("cutting the node range (device)");
    if (c->nblk <= 0) {  // empty row range: nothing to build, nothing to launch
        c->nblk = 0;
        c->has_pipe = false;
        c->has_rows = false;
        c->has_slotpar = false;
        c->a_npos = 0;
        c->npos_gen = 0;
        c->has_partition = true;
        return FH_OK;
    }
    if (!cut_done) {   // tighten the accumulator budget to the largest block actually formed
        long long mx = 1;
        for (size_t b = 0; b + 1 < blk.size(); ++b) mx = std::max<long long>(mx, (long long)h_noff[blk[b + 1]] - h_noff[blk[b]]);
        acc = (int)(S * S * mx);
        HIP_TRY(c, c->blk_off.alloc(blk.size()));
        HIP_TRY(c, hipMemcpyAsync(c->blk_off.p, blk.data(), sizeof(unsigned) * blk.size(), hipMemcpyHostToDevice, c->stream));
        for (size_t b = 0; b + 1 < blk.size(); ++b)
            max_m = std::max(max_m, adj_off_hh[blk[b + 1]] - adj_off_hh[blk[b]]);
    }
    // block tables: unique element lists and packed entries (built once per pattern/partition)
    {
        if (max_m >= 65536) return c->fail(FH_UNSUPPORTED, "gather mode: a node block has more than 65535 adjacent entries");
        const size_t tb = sizeof(int) * 3 * (size_t)std::max(1u, max_m);
        if (tb > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "gather mode: node valence too large for the table builder");
        const int nblk = c->nblk;
        DevBuf<unsigned> counts, uoff;
        HIP_TRY(c, c->gt_hdr.alloc((size_t)nblk + 1));
        HIP_TRY(c, counts.alloc((size_t)nblk + 1));
        HIP_TRY(c, uoff.alloc((size_t)nblk + 1));
        HIP_TRY(c, c->gt_ent.alloc((size_t)c->flat_len + 1));
        auto k0 = k_build_gather_tables<0>;
        auto k1 = k_build_gather_tables<1>;
        // blocks of the usual size: one wavefront per block, four blocks per workgroup (see the kernel)
        const bool by_wave = max_m <= 1024;
        const int wstride = 3 * (int)std::max(1u, max_m);
        if (by_wave) {
            k0 = k_build_gather_tables<0, 4>;
            k1 = k_build_gather_tables<1, 4>;
        }
        const size_t tb_l = by_wave ? 4 * tb : tb;
        const unsigned g_tab = by_wave ? (unsigned)((nblk + 3) / 4) : (unsigned)nblk;
        if (!by_wave && tb > 48 * 1024) {
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb));
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb));
        }
        c->has_pos = max_row < 256;
        if (c->has_pos) HIP_TRY(c, c->gt_pos.alloc((size_t)c->flat_len * c->ei.n + 4));
        hipLaunchKernelGGL(k0, dim3(g_tab), dim3(256), tb_l, c->stream, c->blk_off.p, noff_d, adj_off_d, adj_d, c->ei.n,
                           c->gt_hdr.p, (const unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                           (const unsigned*)nullptr, (unsigned char*)nullptr, nblk, wstride);
        hipLaunchKernelGGL(k_hdr_counts, dim3((nblk + 256) / 256), dim3(256), 0, c->stream, c->gt_hdr.p, nblk, counts.p);
        size_t tmpb = 0;
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tmpb, counts.p, uoff.p, nblk + 1, c->stream));
        DevBuf<char> tmp;
        HIP_TRY(c, tmp.alloc(tmpb + 16));
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tmpb, counts.p, uoff.p, nblk + 1, c->stream));
        unsigned total_u = 0;
        HIP_TRY(c, hipMemcpyAsync(&total_u, uoff.p + nblk, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, c->gt_elems.alloc((size_t)total_u + 1));
        hipLaunchKernelGGL(k1, dim3(g_tab), dim3(256), tb_l, c->stream, c->blk_off.p, noff_d, adj_off_d, adj_d, c->ei.n,
                           c->gt_hdr.p, uoff.p, c->gt_elems.p, c->gt_ent.p, c->conn.p, ncols_d,
                           c->has_pos ? c->gt_pos.p : (unsigned char*)nullptr, nblk, wstride);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    mark("block tables (k_build_gather_tables)");
    // staging capacity: all unique elements of the largest block if that fits the LDS budget
    // (round 5: the two maxima by a device reduction -- the headers themselves, 47 MB for the 216^3 mesh, used to travel to the host for them)
    int umax = 1, mmax = 1, nrow_max = 1;
    if (c->nblk) {
        DevBuf<int> mx;
        HIP_TRY(c, mx.alloc(3));
        const int one[3] = {1, 1, 1};
        HIP_TRY(c, hipMemcpyAsync(mx.p, one, sizeof one, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_hdr_max, dim3((unsigned)std::min(4096, (c->nblk + 255) / 256)), dim3(256), 0, c->stream, c->gt_hdr.p, c->nblk, mx.p);
        HIP_TRY(c, hipGetLastError());
        int got[3] = {1, 1, 1};
        HIP_TRY(c, hipMemcpyAsync(got, mx.p, sizeof got, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        umax = std::max(1, got[0]);
        mmax = std::max(1, got[1]);
        nrow_max = std::max(1, got[2]);
        (void)nrow_max;
    }
    int ub = 0;
    if (layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, umax, acc, 64, true, mb, c->fast_ok) <= lds_target) {
        ub = umax;
    } else {
        for (int t = 1; t <= umax; ++t) {
            const size_t b = layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, t, acc, 64, true, mb, c->fast_ok);
            if (b <= lds_target) ub = t; else break;
        }
    }
    if (ub == 0) {
        ub = 1;
        if (layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, 1, acc, 64, true, mb, c->fast_ok) > LDS_LIMIT)
            return c->fail(FH_UNSUPPORTED, "gather mode: a row block does not fit in LDS; use FH_SCATTER_ATOMIC");
    }
    if (c->env("FENRIS_HIP_VERBOSE"))
        std::fprintf(stderr, "[fenris_hip] gather partition: nblk=%d nb=%d umax=%d mmax=%d acc=%d ub=%d lds=%zu B\n", c->nblk,
                     nb_target, umax, mmax, acc, ub,
                     layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, ub, acc, 64, true, mb, c->fast_ok));
    // position-indexed tables for the pipelined kernel (elements with few geometry nodes, pos table present)
    c->has_pipe = false;
    c->has_rows = false;
    c->a_npos = 0;
    c->npos_gen = c->nblk;
    if (c->has_pos && !c->env("FENRIS_HIP_NO_PIPE") && c->ei.n == c->ei.ng && c->ei.n <= 8 && c->nblk > 0) {
        const int n = c->ei.n;
        const int ms = (mmax + 3) / 4 * 4;
        const int us = (umax + 3) / 4 * 4;
        // local nodes per lane in the pipelined kernel's phase C
        int jt = c->env_int("FENRIS_HIP_PIPE_JT", (n % 2 == 0) ? 2 : n);
        if (jt != 1 && jt != 2 && jt != 4 && jt != n) jt = 1;
        if (n % jt != 0) jt = 1;
        c->p_jt = jt;
        if (us * c->ei.ng <= (rows_special ? 1024 : 512) && us <= 252 && ms <= (rows_special ? 352 : 256) && (rows_special || (ms * (n / jt) <= 256 && ms * n / 4 <= 256)) && ms <= mb &&
            nb_target <= 254 && pipe_record_words(us, ms, n, nb_target) <= (rows_special ? 1024 : 512) &&
            (c->fast_ok || (c->op == FH_MASS_SCALAR && c->elem_kind == FH_HEX8))) {   // (the mass tables take densities that differ from point to point)
            const int nblk = c->nblk;
            mark("headers to the host, staging sizes");
            // Block classes: 1 = every adjacent element is affine, the block runs on k_affine_rows; 0 = general kernels.
            // Chains never mix classes, so each class gets its own sweep order and its own position-indexed tables.
            hvec<unsigned char> cls((size_t)nblk, 0);
            DevBuf<unsigned char> cls_d;
            const bool want_aff = c->elem_kind == FH_HEX8 && c->has_aff && c->has_ghat && c->num_aff > 0 && !c->has_rules &&
                                  !c->aff_failed && c->affine_tol > 0.0 &&
                                  (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC || (c->op == FH_MASS_SCALAR && c->has_params)) &&
                                  us <= 32 && nb_target <= 8 && !c->env("FENRIS_HIP_NO_AFFINE");
            if (want_aff) {
                HIP_TRY(c, cls_d.alloc((size_t)nblk));
                hipLaunchKernelGGL(k_block_class, dim3((nblk + 255) / 256), dim3(256), 0, c->stream, c->gt_hdr.p, c->gt_elems.p,
                                   c->elem_aff.p, nblk, 32, cls_d.p);
                HIP_TRY(c, hipGetLastError());
                HIP_TRY(c, hipMemcpyAsync(cls.data(), cls_d.p, (size_t)nblk, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            }
            mark("block classes");
            // sweep order: chains of blocks whose consecutive members share elements (their staged data is reused)
            hvec<int> order[2], chain_off[2];
            chain_off[0].push_back(0);
            chain_off[1].push_back(0);
            // (every block affine -- structured boxes: no chains to form, the affine positions are sorted into CSR order below)
            bool all_affine = want_aff && std::find(cls.begin(), cls.end(), (unsigned char)0) == cls.end();
            if (c->op == FH_MASS_SCALAR && want_aff && !all_affine) {
                // the mass matrix has no kernel for the general positions alone (the generic gather walks every block): a mesh with
                // any non-affine block stays on it entirely
                std::fill(cls.begin(), cls.end(), (unsigned char)0);
                if (cls_d.p) HIP_TRY(c, hipMemsetAsync(cls_d.p, 0, (size_t)nblk, c->stream));
            }
            // (round 5: ... and no host vectors either -- order and chain offsets of that case are one iota array made on the device; the two
            // host loops with their uploads were 18 ms of the 216^3 mesh's first assembly)
            const bool ident_aff = all_affine;
            if (ident_aff) {
            } else if (!c->env("FENRIS_HIP_NO_SWEEP") && !all_affine) {
                DevBuf<int> node2blk, succ_d;
                HIP_TRY(c, node2blk.alloc((size_t)N + 1));
                HIP_TRY(c, hipMemsetAsync(node2blk.p, 0xff, sizeof(int) * ((size_t)N + 1), c->stream));  // -1: not in a block
                HIP_TRY(c, succ_d.alloc((size_t)nblk));
                hipLaunchKernelGGL(k_node_to_block, dim3((nblk + 255) / 256), dim3(256), 0, c->stream, c->blk_off.p, nblk, node2blk.p);
                if (umax * n <= 256)
                    hipLaunchKernelGGL(k_block_successor<512>, dim3(nblk), dim3(64), 0, c->stream, c->gt_hdr.p, c->gt_elems.p, c->conn.p, n,
                                       node2blk.p, nblk, want_aff ? cls_d.p : (const unsigned char*)nullptr, succ_d.p,
                                       c->part_perm ? r2v_d.p : (const int*)nullptr);
                else
                    hipLaunchKernelGGL(k_block_successor<2048>, dim3(nblk), dim3(64), 0, c->stream, c->gt_hdr.p, c->gt_elems.p, c->conn.p, n,
                                       node2blk.p, nblk, want_aff ? cls_d.p : (const unsigned char*)nullptr, succ_d.p,
                                       c->part_perm ? r2v_d.p : (const int*)nullptr);
                HostBuf<int> succ((size_t)nblk);
                HIP_TRY(c, hipMemcpyAsync(succ.data(), succ_d.p, sizeof(int) * (size_t)nblk, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                hvec<unsigned char> visited((size_t)nblk, 0);
                for (int b = 0; b < nblk; ++b) {
                    if (visited[b]) continue;
                    const int k = cls[b];
                    for (int cur = b; cur >= 0 && cur < nblk && !visited[cur] && cls[cur] == k; cur = succ[cur]) {
                        visited[cur] = 1;
                        order[k].push_back(cur);
                    }
                    chain_off[k].push_back((int)order[k].size());
                }
            } else {
                for (int b = 0; b < nblk; ++b) { order[cls[b]].push_back(b); chain_off[cls[b]].push_back((int)order[cls[b]].size()); }
            }
            mark("successors and chains");
            c->p_cs = us * c->ei.ng;
            c->p_ms = ms;
            c->p_nbs = nb_target;
            c->p_us = us;
            c->p_rw = pipe_record_words(us, ms, n, nb_target);
            // position-indexed tables of one class
            auto build_set_dev = [&](int npos, int nchains, const int* order_p, const int* chain_p, DevBuf<int>& rec, DevBuf<int>& conn,
                                     DevBuf<int>& elem, int by_parity) -> int {
                struct { const int* p; } order_d{order_p}, chain_d{chain_p};
                HIP_TRY(c, rec.alloc((size_t)npos * c->p_rw));
                HIP_TRY(c, conn.alloc((size_t)npos * c->p_cs));
                HIP_TRY(c, elem.alloc((size_t)npos * us));
#define PT_LAUNCH(NGV)                                                                                                           \
    hipLaunchKernelGGL(k_build_pipe_tables<NGV>, dim3(nchains), dim3(64), 0, c->stream, order_d.p, chain_d.p, c->gt_hdr.p,        \
                       c->gt_elems.p, c->gt_ent.p, c->gt_pos.p, noff_d, c->conn.p, n, c->p_cs, ms, nb_target, us, c->p_rw,     \
                       rec.p, conn.p, elem.p, by_parity)
                switch (c->ei.ng) {
                    case 3: PT_LAUNCH(3); break;
                    case 4: PT_LAUNCH(4); break;
                    case 8: PT_LAUNCH(8); break;
                    default: break;
                }
#undef PT_LAUNCH
                HIP_TRY(c, hipGetLastError());
                HIP_TRY(c, hipStreamSynchronize(c->stream));  // the callers' order / chain arrays are released after the return
                return FH_OK;
            };
            auto build_set = [&](const hvec<int>& ord, const hvec<int>& choff, DevBuf<int>& rec, DevBuf<int>& conn,
                                 DevBuf<int>& elem, int by_parity) -> int {
                DevBuf<int> order_d, chain_d;
                HIP_TRY(c, order_d.alloc(ord.size()));
                HIP_TRY(c, chain_d.alloc(choff.size()));
                HIP_TRY(c, hipMemcpyAsync(order_d.p, ord.data(), sizeof(int) * ord.size(), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipMemcpyAsync(chain_d.p, choff.data(), sizeof(int) * choff.size(), hipMemcpyHostToDevice, c->stream));
                return build_set_dev((int)ord.size(), (int)choff.size() - 1, order_d.p, chain_d.p, rec, conn, elem, by_parity);
            };
            c->a_npos = 0;
            if (ident_aff || !order[1].empty()) {
                // the affine kernel keeps nothing staged from one block to the next, and its write-out carries incomplete
                // 128-byte lines from a block to its successor in memory: positions in CSR order, every position its own chain
                DevBuf<int> tmp_rec;  // the pipelined kernel's records: input of the lane builder only
                int rs = FH_OK;
                if (ident_aff) {
                    DevBuf<int> iota_d;   // order[k] = k, chain_off[k] = k
                    HIP_TRY(c, iota_d.alloc((size_t)nblk + 1));
                    hipLaunchKernelGGL(k_iota_int, dim3((unsigned)std::min(4096, (nblk + 256) / 256)), dim3(256), 0, c->stream, iota_d.p, nblk + 1);
                    HIP_TRY(c, hipGetLastError());
                    rs = build_set_dev(nblk, nblk, iota_d.p, iota_d.p, tmp_rec, c->a_conn, c->a_elem, 0);
                } else {
                    std::sort(order[1].begin(), order[1].end());
                    chain_off[1].resize(order[1].size() + 1);
                    for (size_t k = 0; k <= order[1].size(); ++k) chain_off[1][k] = (int)k;
                    rs = build_set(order[1], chain_off[1], tmp_rec, c->a_conn, c->a_elem, 0);
                }
                if (rs) return rs;
                mark("position tables of the affine class (k_build_pipe_tables)");
                const int npos = ident_aff ? nblk : (int)order[1].size();
                c->a_us = us;
                bool bad = false;
                rs = build_lane_tables(c, tmp_rec.p, us, ms, nb_target, npos, S, c->a_conn.p, c->a_elem.p, c->a_hdr, c->a_lanes, c->a_ntab,
                                       c->a_incomplete, bad, "affine rows");
                if (rs) return rs;
                mark("lane tables of the affine class");
                if (bad) {  // a block the lane tables cannot express: everything on the general kernels
                    c->aff_failed = true;
                    return build_partition(c);
                }
                // (The lane tuner of k_hex8_rows applied to these tables -- element records 80 bytes apart, reference blocks -- was measured:
                // headline 4.91 - 5.06 -> 5.15 - 5.30 ms with the read model alone, +-0 with the staging stores in the model, C2 +4 %.  This
                // kernel is not bound by its LDS reads; the builder's order stays.)
                c->a_conn.release();  // input of the lane builder only
                c->a_npos = npos;
                {   // the element range behind these positions (a node range of a few rows -- the interface plane sent first in a
                    // partition -- needs the records of two element layers, not of ten million elements)
                    DevBuf<int> mm;
                    HIP_TRY(c, mm.alloc(2));
                    const int init[2] = {0x7fffffff, -1};
                    HIP_TRY(c, hipMemcpyAsync(mm.p, init, sizeof init, hipMemcpyHostToDevice, c->stream));
                    const size_t cnt = (size_t)npos * us;
                    hipLaunchKernelGGL(k_minmax_nonneg, dim3((unsigned)std::min<size_t>((cnt + 255) / 256, 4096)), dim3(256), 0, c->stream, c->a_elem.p, cnt, mm.p);
                    int got[2] = {0, -1};
                    HIP_TRY(c, hipMemcpyAsync(got, mm.p, sizeof got, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    c->a_emin = got[1] >= 0 ? got[0] : 0;
                    c->a_emax = got[1];
                }
                mark("element range of the affine class");
            }
            c->npos_gen = (int)order[0].size();
            if (!order[0].empty()) {
                int rs = build_set(order[0], chain_off[0], c->p_rec, c->p_conn, c->p_elem, hrows_cand ? 1 : 0);
                if (rs) return rs;
                mark("position tables of the general class");
            }
            c->has_pipe = true;
            if (c->env("FENRIS_HIP_VERBOSE"))
                std::fprintf(stderr, "[fenris_hip] sweep order: %d general blocks in %d chains, %d affine blocks in %d chains (us=%d ms=%d)\n",
                             c->npos_gen, (int)chain_off[0].size() - 1, c->a_npos, ident_aff ? c->a_npos : (int)chain_off[1].size() - 1, us, ms);
            c->has_rows = false;
            const int npg = c->npos_gen;
            // Hex8, Laplace / uniform LinearElastic: lane tables for the general positions as well (k_hex8_rows, hex8_rows.hip: row-owner
            // lanes instead of LDS atomics; the eight-point rule -- checked at the launch).  Under an element mask a block without an active
            // element gets a lane that stores zeros; when the lanes do not suffice for that somewhere, the pipelined kernel stays.  Its
            // tables are kept: they serve every other rule and per-element parameters.
            c->has_hrows = false;
            if (c->elem_kind == FH_HEX8 && (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC) && us <= HEX8_ROWS_US && nb_target <= 8 && npg > 0 &&
                !c->has_rules && !c->env("FENRIS_HIP_NO_HEX8_ROWS")) {
                bool bad = false;
                int rs = build_lane_tables(c, c->p_rec.p, us, ms, nb_target, npg, S, c->p_conn.p, c->p_elem.p, c->h_hdr, c->h_lanes, c->h_ntab,
                                           c->h_incomplete, bad, "hex8 rows", 1);
                if (rs) return rs;
                if (!bad && !c->h_incomplete) {
                    // lanes rearranged so that the sixteen lanes the LDS serves together read different banks (host, unique tables only).
                    // Round 5: not here but in front of the SECOND launch of the kernel (FENRIS_HIP_TUNE_AFTER launches, default 1; 0: here) --
                    // the tuner takes 18 ms on the 216^3 mesh and buys 0.16 ms per assembly (the matrix is the same bit for bit either way): a
                    // caller who assembles once never needs it, a Newton loop pays it on its second assembly.
                    c->h_tune_pending = 0;
                    if (c->h_ntab <= 4096 && !c->env("FENRIS_HIP_NO_LANE_TUNING")) {
                        c->h_tune_pending = 1 + std::max(0, c->env_int("FENRIS_HIP_TUNE_AFTER", 1));
                        if (c->h_tune_pending == 1) { const int rt = hex8_tune_lanes_now(c); if (rt) return rt; }
                    }
                    HIP_TRY(c, c->h_pos.alloc((size_t)npg * 4));
                    HIP_TRY(c, hex8_rows_positions(c->stream, c->p_rec.p, c->p_rw, us, ms, c->h_hdr.p, npg, c->h_pos.p));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    c->has_hrows = true;
                }
                c->h_hdr.release();   // folded into the position records
                mark("lane tables of the general class (hex8 rows)");
                if (c->env("FENRIS_HIP_VERBOSE"))
                    std::fprintf(stderr, "[fenris_hip] row-owner lanes (Hex8, general positions): %s\n", c->has_hrows ? "built" : "mesh not expressible, pipelined kernel kept");
            }
            // Tet4 with a one-point rule: the row-owner kernel is the default (C3: 1.31 -> 0.85 ms), FENRIS_HIP_NO_ROWS keeps
            // the pipelined kernel
            if (c->elem_kind == FH_TET4 && us * 4 <= 1024 && nb_target <= 16 && npg > 0 && !c->env("FENRIS_HIP_NO_ROWS")) {
                c->r_rw = 8 + us / 4 + nb_target + 1 + nb_target;
                DevBuf<unsigned> row_real;   // first entry of every node's real row, in the order of the blocks
                HIP_TRY(c, row_real.alloc((size_t)N + 1));
                hipLaunchKernelGGL(k_row_starts, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->noff.p,
                                   c->part_perm ? v2r_d.p : (const unsigned*)nullptr, N, row_real.p);
                DevBuf<int> st;
                HIP_TRY(c, st.alloc(2));
                HIP_TRY(c, c->r_rec.alloc((size_t)npg * c->r_rw));
                int bad = 0;
                // the kernel reads these tables at every position and is bound by the bytes it moves (profiles/r06_c3_tables.txt): the strides
                // are the most lanes / distinct vertices any position needs, rounded up to 32 -- counted first, by the builders themselves
                int h_st[2] = {0, 0};
                auto counted = [&](auto&& launch) -> int {   // status[1] of a count-only run of a builder
                    HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
                    launch();
                    HIP_TRY(c, hipGetLastError());
                    HIP_TRY(c, hipMemcpyAsync(h_st, st.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    return FH_OK;
                };
                {
                    const int rc_c = counted([&]() {
                        hipLaunchKernelGGL(k_build_row_lanes_tet4, dim3(npg), dim3(64), 0, c->stream, c->p_rec.p, c->p_rw, us, ms, nb_target, npg,
                                           c->r_rw, c->r_rec.p, (unsigned*)nullptr, 256, st.p, row_real.p, 1);
                    });
                    if (rc_c) return rc_c;
                }
                bad = h_st[0];
                if (bad == 0 && h_st[1] > 256) bad = 1;
                if (bad == 0) {
                    const int ls = std::max(32, (h_st[1] + 31) / 32 * 32);
                    c->r_ls = ls;
                    HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
                    HIP_TRY(c, c->r_lanes4.alloc((size_t)npg * ls * 3));
                    hipLaunchKernelGGL(k_build_row_lanes_tet4, dim3(npg), dim3(64), 0, c->stream, c->p_rec.p, c->p_rw, us, ms,
                                       nb_target, npg, c->r_rw, c->r_rec.p, c->r_lanes4.p, ls, st.p, row_real.p, 0);
                    HIP_TRY(c, hipGetLastError());
                    HIP_TRY(c, hipMemcpyAsync(&bad, st.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                }
                mark("row lanes (Tet4)");
                if (bad == 0) {   // the position's unique vertices and the slot words that index them
                    const int rc_c = counted([&]() {
                        hipLaunchKernelGGL(k_build_row_verts_tet4, dim3(npg), dim3(64), 0, c->stream, c->p_conn.p, us, npg, (int*)nullptr, st.p,
                                           ROWS_TET4_VMAX, 1);
                    });
                    if (rc_c) return rc_c;
                    if (h_st[1] > ROWS_TET4_VMAX) bad = 4;
                    if (bad == 0) {
                        const int vn = std::max(32, (h_st[1] + 31) / 32 * 32);
                        c->r_vn = vn;
                        HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
                        HIP_TRY(c, c->r_vconn.alloc((size_t)npg * (vn + us)));
                        hipLaunchKernelGGL(k_build_row_verts_tet4, dim3(npg), dim3(64), 0, c->stream, c->p_conn.p, us, npg, c->r_vconn.p, st.p, vn, 0);
                        HIP_TRY(c, hipGetLastError());
                        HIP_TRY(c, hipMemcpyAsync(&bad, st.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                        HIP_TRY(c, hipStreamSynchronize(c->stream));
                    }
                }
                c->has_rows = bad == 0;
                HIP_TRY(c, hipStreamSynchronize(c->stream));  // row_real is released at the end of this scope
                mark("row vertices (Tet4)");
                if (c->env("FENRIS_HIP_VERBOSE"))
                    std::fprintf(stderr, "[fenris_hip] row-owner lanes (Tet4, %d lanes and %d vertices per position): %s\n", c->r_ls, c->r_vn,
                                 c->has_rows ? "built" : "mesh not expressible, pipelined kernel kept");
            }
        }
    }
    if (c->part_rows_only && !c->has_rows) {  // these tables serve the row-owner kernel only: smaller blocks, then the standard form
        if (++c->rows_try >= 4) c->perm_failed = true;
        c->part_perm = false;
        c->part_rows_only = false;
        return build_partition(c);
    }
    c->has_slotpar = false;
    if (c->has_rules && c->fast_ok && c->op != FH_LAPLACE && !c->has_pipe) {
        // per-element data without the pipelined tables: the generic kernels need the per-point-coefficient layout
        c->fast_ok = false;
        c->elem_par = false;
        return build_partition(c);
    }
    c->g_ub = ub;
    c->g_umax = umax;
    c->g_mb = mb;
    c->g_acc = acc;
    c->g_nb = 64;
    c->has_partition = true;
    mark("the rest");
    return FH_OK;
}

