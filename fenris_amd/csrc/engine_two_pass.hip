// Two-pass owner-computes assembly: dense element matrices (fp64 MFMA for Hex27), then the row gather
#include "engine_internal.hpp"
#include "hex27_blocks.hpp"
#include "two_pass_kernels.hpp"

namespace {

// first pass on the matrix cores (hex27_blocks.hpp): the elements [w0, w1) of the work list; K_e as its upper node-block triangle
int launch_hex27_blocks(fh_ctx* c, long long w0, long long w1, hipStream_t st) {
    if (w1 <= w0) return FH_OK;
    KArgs a;
    fill_common(c, a);
    a.ke_out = c->ke_dense.p;
    a.labels = c->has_mask ? c->active_list.p : nullptr;
    a.work_begin = w0;
    a.work_end = w1;
    if (!c->env("FENRIS_HIP_HEX27_NO_LEX")) {
        a.conn = c->tp_conn.p;   // nodes in lexicographic order (assemble_two_pass built the tables)
        a.gref = c->gref_lex.p;
        a.gref_t = c->gref_t_lex.p;
        a.vtx_pack = c->hex27_vtx_pack;
    } else {
        for (int g = 0; g < 8; ++g) a.vtx_pack |= (unsigned long long)g << (5 * g);
    }
    // (Measured and retired, scripts/attic/hex27_roles_r06.hpp: a wave-specialised form -- four matrix wavefronts + four prologue wavefronts per workgroup,
    // double-buffered operands, one barrier per element -- runs the pass in 3.16 ms against 3.20: the fp64 vector work of the prologue and the matrix
    // instructions share one datapath and add up whichever wavefronts issue them; profiles/r06_c4_triangle.txt section 4.)
    // four workgroups per CU (29.8 KB of LDS and 128 registers each).  The 16 x 16 tile form of rounds 2 - 5 with full planar matrices
    // (scripts/attic/hex27_mfma_tiles_r05.hpp) took 7.1 - 7.4 ms for C4 where this form takes 6.6 - 6.9 (profiles/r06_c4_triangle.txt).
    const size_t lds1 = sizeof(double) * (size_t)Hex27BlkLds::total;
    const int wgs_default = (int)std::min<size_t>(4, LDS_LIMIT / lds1);
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
    // (FENRIS_HIP_TWO_PASS_GRID: tests force many elements / nodes per workgroup on small meshes)
    // eight times the resident workgroups, dispatched in order (each strides over ~24 elements on C4): 1 / 2 / 4 / 16 x the resident ones 6.15 / 6.08 /
    // 6.04 / 6.03 ms, one element per workgroup 6.54 (the tables a workgroup stages once)
    const int grid1 = std::max(1, (int)std::min<long long>(w1 - w0, c->env_int("FENRIS_HIP_TWO_PASS_GRID", 8 * dev_cus * std::max(1, c->env_int("FENRIS_HIP_HEX27_WGS_PER_CU", wgs_default)))));
    void (*kern)(const KArgs, double, double);
    if (a.trace) kern = c->op == FH_NEO_HOOKEAN ? k_hex27_dense_blocks<FH_NEO_HOOKEAN, true> : k_hex27_dense_blocks<FH_LINEAR_ELASTIC, true>;   // FENRIS_HIP_TRACE: per-phase cycles
    else kern = c->op == FH_NEO_HOOKEAN ? k_hex27_dense_blocks<FH_NEO_HOOKEAN> : k_hex27_dense_blocks<FH_LINEAR_ELASTIC>;
    hipLaunchKernelGGL(kern, dim3(grid1), dim3(256), lds1, st, a, c->uni_mu, c->uni_lambda);
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

// second pass: the rows of `count` nodes (node_list, or all nodes in order) from the dense element matrices
template <int SS, typename PT>
int launch_rows_t(fh_ctx* c, int layout, hipStream_t st, const PT* pos, const unsigned* adj_off, const unsigned* adj, double* values_dev,
                  int overwrite, unsigned max_row, const int* node_list, int count, int threads, int grid_cap) {
    if (count <= 0) return FH_OK;
    const int wpb = threads / 64;
    const size_t lds = (size_t)wpb * sizeof(double) * SS * SS * max_row;
    if (layout != 0) {   // node-block triangles: 2 = upper (Hex27 from hex27_blocks.hpp), 1 = lower (the generic first pass with ke_tri)
        if (SS != 3) return c->fail(FH_HIP_ERROR, "two-pass gather: the triangle layouts are for s = 3");
        const int abl = c->env_int("FENRIS_HIP_ABLATE", 0) >> 8;   // (profiling, timing only: bits 8.. = no stores / no value loads / no LDS adds / no clearing in the second pass)
        typedef void (*tri_kernel)(const unsigned*, const unsigned*, const unsigned*, const PT*, const double*, double*, int, int, const int*, int, int, int);
        tri_kernel kt = nullptr;
        const int n = (int)c->ei.n;
        if (layout == 2 && n == 27) kt = abl ? k_rows_from_tri<27, false, PT, true> : k_rows_from_tri<27, false, PT, false>;
        else if (layout == 1 && n == 27) kt = k_rows_from_tri<27, true, PT>;
        else if (layout == 1 && n == 20) kt = k_rows_from_tri<20, true, PT>;
        else if (layout == 1 && n == 10) kt = k_rows_from_tri<10, true, PT>;
        else if (layout == 1 && n == 8) kt = k_rows_from_tri<8, true, PT>;
        else if (layout == 1 && n == 4) kt = k_rows_from_tri<4, true, PT>;
        if (!kt) return c->fail(FH_HIP_ERROR, "two-pass gather: no triangle kernel for this element");
        overwrite = (overwrite ? 1 : 0) | (abl << 8);
        if (lds > 48 * 1024) HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // consecutive nodes per wavefront: 4, more when the grid would pass its cap (small grids in the tests: FENRIS_HIP_TWO_PASS_GRID);
        // nodes per wavefront slot of an XCD's chunk: 1 024 (a chunk = 4 096 nodes; 0 = workgroups in launch order)
        int npw = std::max(1, c->env_int("FENRIS_HIP_TWO_PASS_NODES_PER_WAVE", 4));
        while ((long long)grid_cap * wpb * npw < count) npw *= 2;
        const int grid = std::max(1, (count + wpb * npw - 1) / (wpb * npw));
        hipLaunchKernelGGL(kt, dim3(grid), dim3(threads), lds, st, c->noff.p, adj_off, adj, pos, c->ke_dense.p, values_dev, overwrite, (int)max_row,
                           node_list, count, npw, std::max(0, c->env_int("FENRIS_HIP_TWO_PASS_XCD_CHUNK", 1024) / npw));
        HIP_TRY(c, hipGetLastError());
        return FH_OK;
    }
    void (*kern)(int, int, const unsigned*, const unsigned*, const unsigned*, const PT*, const double*, double*, int, int, const int*, int) =
        k_rows_from_dense<SS, PT>;
    {
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

int launch_rows(fh_ctx* c, int layout, hipStream_t st, const unsigned* adj_off, const unsigned* adj, double* values_dev, int overwrite,
                unsigned max_row, const int* node_list, int count, int threads, int grid_cap) {
    const int S = c->S();
    const bool wide = max_row >= 256;
#define ROWS(SS)                                                                                                                             \
    (wide ? launch_rows_t<SS, unsigned short>(c, layout, st, c->tp_pos16.p, adj_off, adj, values_dev, overwrite, max_row, node_list, count, threads, grid_cap) \
          : launch_rows_t<SS, unsigned char>(c, layout, st, c->tp_pos8.p, adj_off, adj, values_dev, overwrite, max_row, node_list, count, threads, grid_cap))
    if (S == 1) return ROWS(1);
    if (S == 2) return ROWS(2);
    return ROWS(3);
#undef ROWS
}

}  // namespace

// layout of the element matrices between the passes: 0 column-major full matrices (generic first pass), 2 upper node-block triangles
// (hex27_blocks.hpp: Hex27 LinearElastic / NeoHookean with a uniform table of 27 points run on the matrix cores), 1 lower node-block triangles
// (generic first pass, s = 3 on the 3D elements, symmetric operators: half the bytes between the passes, written as runs of 72 bytes)
static int two_pass_layout(fh_ctx* c) {
    const bool mfma = c->elem_kind == FH_HEX27 && (c->op == FH_LINEAR_ELASTIC || c->op == FH_NEO_HOOKEAN) && !c->has_rules &&
                      c->nq == 27 && c->has_params && c->gref_t.p != nullptr && c->has_hex27_perm && !c->env("FENRIS_HIP_NO_MFMA");
    if (mfma) return 2;
    const int n = (int)c->ei.n;
    const bool tri = c->S() == 3 && c->ei.d == 3 && !c->ragged && (n == 4 || n == 8 || n == 10 || n == 20 || n == 27) &&
                     !(c->op == FH_TENSOR && !c->tensor_sym) && !c->env("FENRIS_HIP_TWO_PASS_FULL");
    return tri ? 1 : 0;
}
size_t two_pass_dense_doubles(fh_ctx* c) {
    const size_t ld = (size_t)c->S() * c->ei.n;
    return two_pass_layout(c) != 0 ? (size_t)(c->ei.n * (c->ei.n + 1) / 2) * 9 * c->E : ld * ld * c->E;
}

// Owner-computes for high-order elements (n > 8), two passes: dense element matrices (element-parallel, every K_e
// computed once), then one wavefront per node gathers the columns of its elements' K_e into its CSR rows.
// Recomputing K_e per owning node block, as the one-pass kernels do, costs 8-27x for a 27-node element.
int assemble_two_pass(fh_ctx* c, double* values_dev, int overwrite) {
    const int S = c->S();
    const int layout = two_pass_layout(c);
    const bool mfma = layout == 2;
    const size_t ke_doubles = two_pass_dense_doubles(c);
    if (c->ke_dense.n < ke_doubles) HIP_TRY(c, c->ke_dense.alloc(ke_doubles));
    const unsigned max_row = c->max_row;  // longest node row, cached with the pattern (no O(N) host scan per assembly)
    if ((size_t)4 * sizeof(double) * S * S * max_row > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "two-pass gather: a node row does not fit in LDS");
    if (max_row >= 65536) return c->fail(FH_UNSUPPORTED, "two-pass gather: node valence too large");
    const unsigned* adj_off = c->has_mask ? c->n2e_off_c.p : c->n2e_off.p;
    const unsigned* adj = c->has_mask ? c->n2e_c.p : c->n2e.p;
    const bool wide = max_row >= 256;
    const bool lex = layout == 2 && !c->env("FENRIS_HIP_HEX27_NO_LEX");
    if (!c->has_tp_pos || c->tp_pos_layout != layout + (lex ? 16 : 0)) {  // once per pattern / element mask / layout
        // number of (node, element) adjacencies: the last offset, read from the device (round 5: this used to pull both offset arrays to the host)
        unsigned last_off = 0;
        HIP_TRY(c, hipMemcpyAsync(&last_off, adj_off + c->N, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const long long entries = (long long)last_off;
        const int* conn_tab = c->conn.p;
        const unsigned* adj_tab = adj;
        if (lex) {   // the triangle layout works on the nodes in lexicographic order (engine_internal.hpp: hex27_perm)
            DevBuf<int> perm;
            HIP_TRY(c, perm.alloc(27));
            HIP_TRY(c, hipMemcpyAsync(perm.p, c->hex27_perm, sizeof(int) * 27, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, c->tp_conn.alloc((size_t)c->E * 27));
            HIP_TRY(c, c->tp_adj.alloc((size_t)entries + 1));
            const long long tc = (long long)c->E * 27;
            hipLaunchKernelGGL(k_permute_conn27, dim3((unsigned)((tc + 255) / 256)), dim3(256), 0, c->stream, tc, c->conn.p, perm.p, c->tp_conn.p);
            if (entries) hipLaunchKernelGGL(k_permute_entries27, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, c->stream, entries, adj, perm.p, c->tp_adj.p);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(c->stream));   // perm is released on scope exit
            conn_tab = c->tp_conn.p;
            adj_tab = c->tp_adj.p;
        } else {
            c->tp_conn.release();
            c->tp_adj.release();
        }
        DevBuf<int> entry_node;
        HIP_TRY(c, entry_node.alloc((size_t)entries + 1));
        hipLaunchKernelGGL(k_entry_nodes, dim3(((int)c->N + 255) / 256), dim3(256), 0, c->stream, (int)c->N, adj_off, entry_node.p);
        const long long total = entries * c->ei.n;
        const int g = (int)((total + 255) / 256);
        if (wide) {
            HIP_TRY(c, c->tp_pos16.alloc((size_t)total + 1));
            if (total) hipLaunchKernelGGL((k_entry_positions<unsigned short>), dim3(g), dim3(256), 0, c->stream, total, c->ei.n, adj_off, adj_tab,
                                          c->noff.p, c->ncols.p, conn_tab, entry_node.p, c->tp_pos16.p);
        } else {
            HIP_TRY(c, c->tp_pos8.alloc((size_t)total + 1));
            if (total) hipLaunchKernelGGL((k_entry_positions<unsigned char>), dim3(g), dim3(256), 0, c->stream, total, c->ei.n, adj_off, adj_tab,
                                          c->noff.p, c->ncols.p, conn_tab, entry_node.p, c->tp_pos8.p);
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // entry_node is released on scope exit
        c->has_tp_pos = true;
        c->tp_pos_layout = layout + (lex ? 16 : 0);
    }
    // (Triangle layout, measured and not kept -- profiles/r06_c4_triangle.txt: walking the nodes in Morton order of their coordinates, by their
    // first adjacent element, and / or giving every XCD a contiguous part of the node list.  Morton + XCD parts halve the reads that leave the
    // L2s (17.7 -> 10.1 GB) and make the pass SLOWER, 3.4 -> 3.6 - 4.1 ms; the pass follows its occupancy instead (4 / 8 / 16 / 20 wavefronts per
    // CU: 7.4 / 4.5 / 3.4 / ~3.25 ms), which registers and the LDS rows of the longest node row hold at 16.)
    const int* node_list = nullptr;
    const int rows_grid_cap = c->env_int("FENRIS_HIP_TWO_PASS_ROWS_GRID", c->env_int("FENRIS_HIP_TWO_PASS_GRID", 1 << 17));   // (C4: 2^17 workgroups 8.33 ms, one per four nodes (410 k) 8.42, 2^13 8.45, 2^11 8.68)
    c->last_kernel = mfma ? "k_hex27_dense_blocks + k_rows_from_tri" : layout == 1 ? "k_assemble_matrix<dump> + k_rows_from_tri" : "k_assemble_matrix<dump> + k_rows_from_dense";

    // ---- serial form: all element matrices, then all rows
    if (mfma) {
        const int r1 = launch_hex27_blocks(c, 0, (long long)(c->has_mask ? c->num_active : c->E), c->stream);
        if (r1) return r1;
    } else {
        const int r1 = element_matrices_enqueue(c, 0, c->has_mask ? c->num_active : c->E, c->ke_dense.p, true, layout == 1);
        if (r1) return r1;
    }
    return launch_rows(c, layout, c->stream, adj_off, lex ? c->tp_adj.p : adj, values_dev, overwrite, max_row, node_list, (int)c->N, 256, rows_grid_cap);
}
