// Two-pass owner-computes assembly: dense element matrices (fp64 MFMA for Hex27), then the row gather
#include "engine_internal.hpp"
#include "hex27_mfma.hpp"
#include "hex27_blocks.hpp"
#include "two_pass_kernels.hpp"

namespace {

// first pass on the matrix cores (hex27_mfma.hpp): the elements [w0, w1) of the work list, planar layout
int launch_hex27_mfma(fh_ctx* c, long long w0, long long w1, hipStream_t st) {
    if (w1 <= w0) return FH_OK;
    KArgs a;
    fill_common(c, a);
    a.ke_out = c->ke_dense.p;
    a.labels = c->has_mask ? c->active_list.p : nullptr;
    a.work_begin = w0;
    a.work_end = w1;
    // FENRIS_HIP_HEX27_FORM: 2 (default, round 6) = 4 x 4 x 4 blocks with one operand array, four workgroups per CU (hex27_blocks.hpp);
    // 0 = the 16 x 16 tiles, 1 = the round-5 block experiment (hex27_mfma.hpp)
    const int form = c->env_int("FENRIS_HIP_HEX27_FORM", c->env_int("FENRIS_HIP_HEX27_BLOCKS", 0) != 0 ? 1 : 2);
    const bool blocks2 = form == 2 && c->gref_t.p != nullptr;
    const size_t lds1 = sizeof(double) * (size_t)(blocks2 ? Hex27BlkLds::total : Hex27Lds::total);
    const int wgs_default = (int)std::min<size_t>(blocks2 ? 4 : 3, LDS_LIMIT / lds1);
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
    // (FENRIS_HIP_TWO_PASS_GRID: tests force many elements / nodes per workgroup on small meshes)
    if (st == c->tp_stream1 && c->tp_gather_cus > 0) dev_cus = std::max(1, dev_cus - c->tp_gather_cus);   // (CU-masked stream: the CUs left to this pass)
    const int grid1 = std::max(1, (int)std::min<long long>(w1 - w0, c->env_int("FENRIS_HIP_TWO_PASS_GRID", dev_cus * std::max(1, c->env_int("FENRIS_HIP_HEX27_WGS_PER_CU", wgs_default)))));
    if (blocks2) {
        void (*kern)(const KArgs, double, double);
        if (a.trace) kern = c->op == FH_NEO_HOOKEAN ? k_hex27_dense_blocks<FH_NEO_HOOKEAN, true> : k_hex27_dense_blocks<FH_LINEAR_ELASTIC, true>;
        else kern = c->op == FH_NEO_HOOKEAN ? k_hex27_dense_blocks<FH_NEO_HOOKEAN> : k_hex27_dense_blocks<FH_LINEAR_ELASTIC>;
        hipLaunchKernelGGL(kern, dim3(grid1), dim3(256), lds1, st, a, c->uni_mu, c->uni_lambda);
        HIP_TRY(c, hipGetLastError());
        return FH_OK;
    }
    if (c->op == FH_NEO_HOOKEAN && a.trace) {   // FENRIS_HIP_TRACE: per-phase cycle counters
        auto kern = form == 1 ? k_hex27_dense_mfma<FH_NEO_HOOKEAN, true, 1> : k_hex27_dense_mfma<FH_NEO_HOOKEAN, true>;
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(kern, dim3(grid1), dim3(256), lds1, st, a, c->uni_mu, c->uni_lambda);
    } else if (form == 1) {
        // the 4 x 4 x 4 block form (round 5 experiment, hex27_mfma.hpp "second form"): parity-green and bit-symmetric; half the matrix-core time
        // of the 16 x 16 tiles and still 3 - 4 % SLOWER end to end (7.30 - 7.38 against 7.04 - 7.07 ms on one box: the pass is latency-bound once
        // the matrix instructions shrink, profiles/r05_c4_mfma_blocks.txt): opt-in
        auto kern = c->op == FH_NEO_HOOKEAN ? k_hex27_dense_mfma<FH_NEO_HOOKEAN, false, 1> : k_hex27_dense_mfma<FH_LINEAR_ELASTIC, false, 1>;
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(kern, dim3(grid1), dim3(256), lds1, st, a, c->uni_mu, c->uni_lambda);
    } else if (c->op == FH_NEO_HOOKEAN) {
        auto kern = k_hex27_dense_mfma<FH_NEO_HOOKEAN>;
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(kern, dim3(grid1), dim3(256), lds1, st, a, c->uni_mu, c->uni_lambda);
    } else {
        auto kern = k_hex27_dense_mfma<FH_LINEAR_ELASTIC>;
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(kern, dim3(grid1), dim3(256), lds1, st, a, c->uni_mu, c->uni_lambda);
    }
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

// second pass: the rows of `count` nodes (node_list, or all nodes in order) from the dense element matrices
template <int SS, typename PT>
int launch_rows_t(fh_ctx* c, bool planar, hipStream_t st, const PT* pos, const unsigned* adj_off, const unsigned* adj, double* values_dev,
                  int overwrite, unsigned max_row, const int* node_list, int count, int threads, int grid_cap) {
    if (count <= 0) return FH_OK;
    const int wpb = threads / 64;
    const size_t lds = (size_t)wpb * sizeof(double) * SS * SS * max_row;
    void (*kern)(int, int, const unsigned*, const unsigned*, const unsigned*, const PT*, const double*, double*, int, int, const int*, int) =
        planar ? k_rows_from_dense<SS, PT, true> : k_rows_from_dense<SS, PT, false>;
    if (!planar && !c->env("FENRIS_HIP_NO_ROWS_SMALL")) {
        const int ld_ = SS * (int)c->ei.n;
        if (ld_ <= 8) kern = k_rows_from_dense_small<SS, PT, 8>;
        else if (ld_ <= 16) kern = k_rows_from_dense_small<SS, PT, 16>;
        else if (ld_ <= 32) kern = k_rows_from_dense_small<SS, PT, 32>;
    }
    if (lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = std::max(1, std::min((count + wpb - 1) / wpb, grid_cap));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, (int)c->N, c->ei.n, c->noff.p, adj_off, adj, pos, c->ke_dense.p, values_dev,
                       overwrite, (int)max_row, node_list, count);
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

int launch_rows(fh_ctx* c, bool planar, hipStream_t st, const unsigned* adj_off, const unsigned* adj, double* values_dev, int overwrite,
                unsigned max_row, const int* node_list, int count, int threads, int grid_cap) {
    const int S = c->S();
    const bool wide = max_row >= 256;
#define ROWS(SS)                                                                                                                             \
    (wide ? launch_rows_t<SS, unsigned short>(c, planar, st, c->tp_pos16.p, adj_off, adj, values_dev, overwrite, max_row, node_list, count, threads, grid_cap) \
          : launch_rows_t<SS, unsigned char>(c, planar, st, c->tp_pos8.p, adj_off, adj, values_dev, overwrite, max_row, node_list, count, threads, grid_cap))
    if (S == 1) return ROWS(1);
    if (S == 2) return ROWS(2);
    return ROWS(3);
#undef ROWS
}

// Overlapped form, once per pattern: the nodes sorted by the chunk of their last adjacent element
int build_chunk_lists(fh_ctx* c, int chunks, const unsigned* adj_off, const unsigned* adj) {
    const int N = (int)c->N;
    DevBuf<unsigned> keys, keys2, ids, ids2, counts;
    HIP_TRY(c, keys.alloc((size_t)N));
    HIP_TRY(c, keys2.alloc((size_t)N));
    HIP_TRY(c, ids.alloc((size_t)N));
    HIP_TRY(c, ids2.alloc((size_t)N));
    HIP_TRY(c, counts.alloc((size_t)chunks));
    HIP_TRY(c, hipMemsetAsync(counts.p, 0, sizeof(unsigned) * (size_t)chunks, c->stream));
    hipLaunchKernelGGL(k_node_last_chunk, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, c->ei.n, adj_off, adj, (long long)c->E, chunks, keys.p,
                       ids.p, counts.p);
    HIP_TRY(c, hipGetLastError());
    // stable radix sort by chunk: node ids stay ascending inside a chunk (their rows are neighbours in memory)
    int bits = 1;
    while ((1 << bits) < chunks) ++bits;
    size_t tmp_bytes = 0;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.p, keys2.p, ids.p, ids2.p, N, 0, bits, c->stream));
    DevBuf<unsigned char> tmp;
    HIP_TRY(c, tmp.alloc(tmp_bytes + 16));
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.p, keys2.p, ids.p, ids2.p, N, 0, bits, c->stream));
    HIP_TRY(c, c->tp_nodes.alloc((size_t)N + 1));
    HIP_TRY(c, hipMemcpyAsync(c->tp_nodes.p, ids2.p, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice, c->stream));
    std::vector<unsigned> h((size_t)chunks);
    HIP_TRY(c, hipMemcpyAsync(h.data(), counts.p, sizeof(unsigned) * (size_t)chunks, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // (also: the temporaries are released on return)
    c->tp_chunk_off.assign((size_t)chunks + 1, 0);
    for (int k = 0; k < chunks; ++k) c->tp_chunk_off[(size_t)k + 1] = c->tp_chunk_off[(size_t)k] + (int)h[(size_t)k];
    if (c->tp_chunk_off[(size_t)chunks] != N) return c->fail(FH_HIP_ERROR, "two-pass gather: chunk lists do not cover the nodes");
    c->tp_chunks = chunks;
    return FH_OK;
}

}  // namespace

// Owner-computes for high-order elements (n > 8), two passes: dense element matrices (element-parallel, every K_e
// computed once), then one wavefront per node gathers the columns of its elements' K_e into its CSR rows.
// Recomputing K_e per owning node block, as the one-pass kernels do, costs 8-27x for a 27-node element.
//
// Round 5, the OVERLAPPED form (Hex27 on the matrix cores, no element mask; FENRIS_HIP_TWO_PASS_CHUNKS, 0 = the serial form): the first pass is
// bound by the matrix cores and the fp64 pipe and leaves most of the HBM bandwidth idle, the second is pure HBM traffic -- so the element
// matrices are written in chunks of consecutive elements on the context's stream, and as soon as chunk k is complete the rows of the nodes
// whose LAST adjacent element lies in chunk k are gathered on a second stream, beside the matrices of chunk k + 1 (a node's rows need all
// its elements: nodes on the border between two chunks wait for the later one).  Same kernels, same per-node order of additions: the
// result is bit for bit that of the serial form.  The context's stream waits for the last gather before anything else runs on it.
int assemble_two_pass(fh_ctx* c, double* values_dev, int overwrite) {
    const int S = c->S();
    const size_t ld = (size_t)S * c->ei.n;
    if (c->ke_dense.n < ld * ld * c->E) HIP_TRY(c, c->ke_dense.alloc(ld * ld * c->E));
    // first pass: Hex27 LinearElastic / NeoHookean with a uniform table run on the matrix cores (hex27_mfma.hpp) and
    // write the planar layout; everything else takes the generic element kernel (column-major K_e)
    const bool mfma = c->elem_kind == FH_HEX27 && (c->op == FH_LINEAR_ELASTIC || c->op == FH_NEO_HOOKEAN) && !c->has_rules &&
                      c->nq == 27 && c->has_params && !c->env("FENRIS_HIP_NO_MFMA");
    const unsigned max_row = c->max_row;  // longest node row, cached with the pattern (no O(N) host scan per assembly)
    if ((size_t)4 * sizeof(double) * S * S * max_row > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "two-pass gather: a node row does not fit in LDS");
    if (max_row >= 65536) return c->fail(FH_UNSUPPORTED, "two-pass gather: node valence too large");
    const unsigned* adj_off = c->has_mask ? c->n2e_off_c.p : c->n2e_off.p;
    const unsigned* adj = c->has_mask ? c->n2e_c.p : c->n2e.p;
    const bool wide = max_row >= 256;
    if (!c->has_tp_pos) {  // once per pattern / element mask
        // number of (node, element) adjacencies: the last offset, read from the device (round 5: this used to pull both offset arrays to the host)
        unsigned last_off = 0;
        HIP_TRY(c, hipMemcpyAsync(&last_off, adj_off + c->N, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const long long entries = (long long)last_off;
        DevBuf<int> entry_node;
        HIP_TRY(c, entry_node.alloc((size_t)entries + 1));
        hipLaunchKernelGGL(k_entry_nodes, dim3(((int)c->N + 255) / 256), dim3(256), 0, c->stream, (int)c->N, adj_off, entry_node.p);
        const long long total = entries * c->ei.n;
        const int g = (int)((total + 255) / 256);
        if (wide) {
            HIP_TRY(c, c->tp_pos16.alloc((size_t)total + 1));
            if (total) hipLaunchKernelGGL((k_entry_positions<unsigned short>), dim3(g), dim3(256), 0, c->stream, total, c->ei.n, adj_off, adj,
                                          c->noff.p, c->ncols.p, c->conn.p, entry_node.p, c->tp_pos16.p);
        } else {
            HIP_TRY(c, c->tp_pos8.alloc((size_t)total + 1));
            if (total) hipLaunchKernelGGL((k_entry_positions<unsigned char>), dim3(g), dim3(256), 0, c->stream, total, c->ei.n, adj_off, adj,
                                          c->noff.p, c->ncols.p, c->conn.p, entry_node.p, c->tp_pos8.p);
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // entry_node is released on scope exit
        c->has_tp_pos = true;
        c->tp_chunks = 0;   // the chunk lists belong to the same pattern / mask
    }
    const int rows_grid_cap = c->env_int("FENRIS_HIP_TWO_PASS_ROWS_GRID", c->env_int("FENRIS_HIP_TWO_PASS_GRID", 1 << 17));   // (C4: 2^17 workgroups 8.33 ms, one per four nodes (410 k) 8.42, 2^13 8.45, 2^11 8.68)
    c->last_kernel = mfma ? "k_hex27_dense_mfma + k_rows_from_dense" : "k_assemble_matrix<dump> + k_rows_from_dense";

    // ---- overlapped form
    int chunks = (mfma && !c->has_mask) ? c->env_int("FENRIS_HIP_TWO_PASS_CHUNKS", 0) : 0;
    if (chunks > 1 && (long long)c->E / chunks < 1) chunks = (int)std::min<long long>(chunks, (long long)c->E);
    if (chunks > 1) {
        chunks = std::min(chunks, 1024);
        if (c->tp_chunks != chunks) { const int rb = build_chunk_lists(c, chunks, adj_off, adj); if (rb) return rb; }
        // FENRIS_HIP_TWO_PASS_GATHER_CUS = g > 0: the two passes on DISJOINT sets of CUs (hipExtStreamCreateWithCUMask) -- every (256 / g)-th CU
        // gathers, the others multiply.  Side by side on the same CUs the two kernels only take turns (profiles/r05_c4_overlap.txt).
        const int gcus = c->env_int("FENRIS_HIP_TWO_PASS_GATHER_CUS", 0);
        if (c->tp_gather_cus != gcus) {
            if (c->tp_stream) { (void)hipStreamDestroy(c->tp_stream); c->tp_stream = nullptr; }
            if (c->tp_stream1) { (void)hipStreamDestroy(c->tp_stream1); c->tp_stream1 = nullptr; }
            int dev_cus = 256;
            (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
            if (gcus > 0 && gcus < dev_cus) {
                const int words = (dev_cus + 31) / 32;
                std::vector<uint32_t> mg((size_t)words, 0u), mp((size_t)words, 0u);
                const int shift = c->env_int("FENRIS_HIP_TWO_PASS_GATHER_CU_SHIFT", 0);
                for (int cu = 0; cu < dev_cus; ++cu) {
                    // CU `cu` gathers when it is one of g evenly spread ones
                    const bool g = ((long long)(cu + shift) * gcus / dev_cus) != ((long long)(cu + shift + 1) * gcus / dev_cus);
                    (g ? mg : mp)[(size_t)cu / 32] |= 1u << (cu % 32);
                }
                HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->tp_stream, (uint32_t)words, mg.data()));
                HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->tp_stream1, (uint32_t)words, mp.data()));
            } else {
                HIP_TRY(c, hipStreamCreateWithFlags(&c->tp_stream, hipStreamNonBlocking));
            }
            c->tp_gather_cus = gcus;
        }
        hipStream_t s1 = c->tp_stream1 ? c->tp_stream1 : c->stream;
        while ((int)c->tp_events.size() < chunks + 3) {
            hipEvent_t ev;
            HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            c->tp_events.push_back(ev);
        }
        const int gthreads = c->env_int("FENRIS_HIP_TWO_PASS_GATHER_THREADS", 256) >= 256 ? 256 : (c->env_int("FENRIS_HIP_TWO_PASS_GATHER_THREADS", 256) >= 128 ? 128 : 64);
        hipEvent_t ev_start = c->tp_events[(size_t)chunks], ev_done = c->tp_events[(size_t)chunks + 1], ev_done1 = c->tp_events[(size_t)chunks + 2];
        // whatever ran on the context's stream before (the previous reader of `values`) is finished before the first gather writes
        HIP_TRY(c, hipEventRecord(ev_start, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->tp_stream, ev_start, 0));
        if (s1 != c->stream) HIP_TRY(c, hipStreamWaitEvent(s1, ev_start, 0));
        for (int k = 0; k < chunks; ++k) {
            const long long e0 = (long long)k * (long long)c->E / chunks, e1 = (long long)(k + 1) * (long long)c->E / chunks;
            const int r1 = launch_hex27_mfma(c, e0, e1, s1);
            if (r1) return r1;
            HIP_TRY(c, hipEventRecord(c->tp_events[(size_t)k], s1));
            HIP_TRY(c, hipStreamWaitEvent(c->tp_stream, c->tp_events[(size_t)k], 0));
            const int n0 = c->tp_chunk_off[(size_t)k], n1 = c->tp_chunk_off[(size_t)k + 1];
            const int r2 = launch_rows(c, true, c->tp_stream, adj_off, adj, values_dev, overwrite, max_row, c->tp_nodes.p + n0, n1 - n0, gthreads, rows_grid_cap);
            if (r2) return r2;
        }
        HIP_TRY(c, hipEventRecord(ev_done, c->tp_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, ev_done, 0));
        if (s1 != c->stream) {
            HIP_TRY(c, hipEventRecord(ev_done1, s1));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, ev_done1, 0));
        }
        return FH_OK;
    }

    // ---- serial form: all element matrices, then all rows
    if (mfma) {
        const int r1 = launch_hex27_mfma(c, 0, (long long)(c->has_mask ? c->num_active : c->E), c->stream);
        if (r1) return r1;
    } else {
        const int r1 = element_matrices_enqueue(c, 0, c->has_mask ? c->num_active : c->E, c->ke_dense.p, true);
        if (r1) return r1;
    }
    return launch_rows(c, mfma, c->stream, adj_off, adj, values_dev, overwrite, max_row, nullptr, (int)c->N, 256, rows_grid_cap);
}
