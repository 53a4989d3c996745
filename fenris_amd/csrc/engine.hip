// Context, C ABI for mesh / operator / quadrature / pattern / colouring / options (see engine_internal.hpp for the split)
#include "engine_internal.hpp"

int grid_for(long long n, int block, int cap) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

void invalidate_pattern(fh_ctx* c) {
    c->has_pattern = false;
    c->has_partition = false; ++c->struct_gen; c->has_tp_pos = false;
    c->has_colors = false;
    c->nnz_nodes = 0;
}

// node -> (active element, local index) adjacency used by the owner-computes kernels when an element
// mask is set (multi-GPU partitions: the pattern comes from own + halo elements, numerics from own ones)
int build_compute_adjacency(fh_ctx* c) {
    c->has_partition = false; ++c->struct_gen; c->has_tp_pos = false;
    if (!c->has_mask || !c->has_pattern) return FH_OK;
    const int N = (int)c->N;
    hipStream_t st = c->stream;
    ConnView cv{c->conn.p, nullptr, nullptr, c->ei.n, (long long)c->flat_len, c->active.p};
    DevBuf<unsigned> deg, cursor;
    DevBuf<int> flags;
    HIP_TRY(c, deg.alloc((size_t)N + 1));
    HIP_TRY(c, cursor.alloc((size_t)N + 1));
    HIP_TRY(c, flags.alloc(2));
    HIP_TRY(c, c->n2e_off_c.alloc((size_t)N + 1));
    HIP_TRY(c, hipMemsetAsync(deg.p, 0, sizeof(unsigned) * ((size_t)N + 1), st));
    HIP_TRY(c, hipMemsetAsync(cursor.p, 0, sizeof(unsigned) * ((size_t)N + 1), st));
    HIP_TRY(c, hipMemsetAsync(flags.p, 0, sizeof(int) * 2, st));
    if (c->flat_len > 0)
        hipLaunchKernelGGL(k_count_degree, dim3(grid_for((long long)c->flat_len, 256)), dim3(256), 0, st, cv, deg.p, N, flags.p);
    size_t tmp_bytes = 0;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, deg.p, c->n2e_off_c.p, N + 1, st));
    DevBuf<char> tmp;
    HIP_TRY(c, tmp.alloc(tmp_bytes + 16));
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, deg.p, c->n2e_off_c.p, N + 1, st));
    HIP_TRY(c, c->n2e_c.alloc((size_t)c->flat_len + 1));
    if (c->flat_len > 0) {
        hipLaunchKernelGGL(k_fill_n2e, dim3(grid_for((long long)c->flat_len, 256)), dim3(256), 0, st, cv, c->n2e_off_c.p, cursor.p,
                           c->n2e_c.p, N);
        hipLaunchKernelGGL(k_sort_n2e, dim3(grid_for(N, 256, 1 << 30)), dim3(256), 0, st, c->n2e_off_c.p, c->n2e_c.p, N);
    }
    c->h_n2e_off_c.assign((size_t)N + 1, 0);
    HIP_TRY(c, hipMemcpyAsync(c->h_n2e_off_c.data(), c->n2e_off_c.p, sizeof(unsigned) * ((size_t)N + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

// node -> (element, local node) adjacency alone (a context that holds a mesh but no operator: ElementSourceAssembler): what
// build_pattern computes first, into buffers of its own
int build_source_adjacency(fh_ctx* c) {
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, "source adjacency: no finite element mesh set");
    if (c->src_adj_gen == c->struct_gen && c->src_n2e_off.p) return FH_OK;
    const int N = (int)c->N;
    hipStream_t st = c->stream;
    ConnView cv{c->conn.p, nullptr, nullptr, c->ei.n, (long long)c->flat_len, nullptr};
    DevBuf<unsigned> deg, cursor;
    DevBuf<int> flags;
    HIP_TRY(c, deg.alloc((size_t)N + 1));
    HIP_TRY(c, cursor.alloc((size_t)N + 1));
    HIP_TRY(c, flags.alloc(2));
    HIP_TRY(c, c->src_n2e_off.alloc((size_t)N + 1));
    HIP_TRY(c, hipMemsetAsync(deg.p, 0, sizeof(unsigned) * ((size_t)N + 1), st));
    HIP_TRY(c, hipMemsetAsync(cursor.p, 0, sizeof(unsigned) * ((size_t)N + 1), st));
    HIP_TRY(c, hipMemsetAsync(flags.p, 0, sizeof(int) * 2, st));
    if (c->flat_len > 0)
        hipLaunchKernelGGL(k_count_degree, dim3(grid_for((long long)c->flat_len, 256)), dim3(256), 0, st, cv, deg.p, N, flags.p);
    size_t tmp_bytes = 0;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, deg.p, c->src_n2e_off.p, N + 1, st));
    DevBuf<char> tmp;
    HIP_TRY(c, tmp.alloc(tmp_bytes + 16));
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, deg.p, c->src_n2e_off.p, N + 1, st));
    HIP_TRY(c, c->src_n2e.alloc((size_t)c->flat_len + 1));
    if (c->flat_len > 0) {
        hipLaunchKernelGGL(k_fill_n2e, dim3(grid_for((long long)c->flat_len, 256)), dim3(256), 0, st, cv, c->src_n2e_off.p, cursor.p,
                           c->src_n2e.p, N);
        hipLaunchKernelGGL(k_sort_n2e, dim3(grid_for(N, 256, 1 << 30)), dim3(256), 0, st, c->src_n2e_off.p, c->src_n2e.p, N);
    }
    int h_flags[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipGetLastError());
    if (h_flags[0]) return c->fail(FH_BAD_ARGUMENT, "connectivity refers to a node index >= num_nodes");
    c->src_adj_gen = c->struct_gen;
    return FH_OK;
}

// host copies of the node-level row offsets and of the node -> element offsets: made when something on the host needs them (the gather
// block partition, fh_pattern's row offsets), not by the pattern build itself (82 MB over PCIe on the 216^3 mesh)
int host_offsets(fh_ctx* c) {
    const size_t N = (size_t)c->N;
    if (c->h_noff.size() == N + 1 && c->h_n2e_off.size() == N + 1) return FH_OK;
    // into temporaries, swapped in on success: a failed copy must not leave vectors of the right SIZE with garbage behind ("already cached")
    std::vector<unsigned> a(N + 1), b(N + 1);
    HIP_TRY(c, hipMemcpyAsync(a.data(), c->noff.p, sizeof(unsigned) * (N + 1), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(b.data(), c->n2e_off.p, sizeof(unsigned) * (N + 1), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->h_noff.swap(a);
    c->h_n2e_off.swap(b);
    return FH_OK;
}

// ---------------------------------------------------------------------------------- pattern build
int build_pattern(fh_ctx* c) {
    if (!c->has_mesh) return c->fail(FH_INVALID_STATE, "fh_pattern: no mesh/connectivity set");
    if (c->S() <= 0) return c->fail(FH_INVALID_STATE, "fh_pattern: no operator set (solution dim unknown)");
    if (c->has_pattern) return FH_OK;
    const int N = (int)c->N;
    hipStream_t st = c->stream;
    ConnView cv{c->conn.p, c->ragged ? c->eoff.p : nullptr, c->ragged ? c->k2e.p : nullptr, c->ei.n, (long long)c->flat_len, nullptr};
    // the temporaries of the build in ONE allocation (degrees, cursors, counts, flags, heavy-node list, reduction results, hipcub's
    // scratch): on the reference's small benchmark meshes (benches/assembly.rs:147-241: 1 500 ... 96 000 tetrahedra) a dozen hipMalloc /
    // hipFree pairs were most of the build
    size_t scan_bytes = 0, red_bytes = 0;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (unsigned*)nullptr, (unsigned*)nullptr, N + 1, st));
    HIP_TRY(c, hipcub::DeviceReduce::Max(nullptr, red_bytes, (unsigned*)nullptr, (unsigned*)nullptr, N + 1, st));
    size_t tmp_bytes = std::max(scan_bytes, red_bytes) + 16;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t n1 = al(sizeof(unsigned) * ((size_t)N + 1)), zeroed = 3 * n1 + al(2 * sizeof(int));
    DevBuf<char> arena;
    HIP_TRY(c, arena.alloc(zeroed + al(sizeof(int) * HEAVY_CAP) + al(4 * sizeof(unsigned)) + al(tmp_bytes)));
    struct { unsigned* p; } deg{reinterpret_cast<unsigned*>(arena.p)}, cursor{reinterpret_cast<unsigned*>(arena.p + n1)},
        cnt{reinterpret_cast<unsigned*>(arena.p + 2 * n1)}, red{reinterpret_cast<unsigned*>(arena.p + zeroed + al(sizeof(int) * HEAVY_CAP))};
    struct { int* p; } flags{reinterpret_cast<int*>(arena.p + 3 * n1)}, heavy{reinterpret_cast<int*>(arena.p + zeroed)};
    struct { char* p; } tmp{arena.p + zeroed + al(sizeof(int) * HEAVY_CAP) + al(4 * sizeof(unsigned))};
    HIP_TRY(c, c->n2e_off.alloc((size_t)N + 1));
    HIP_TRY(c, c->noff.alloc((size_t)N + 1));
    HIP_TRY(c, hipMemsetAsync(arena.p, 0, zeroed, st));
    if (c->flat_len > 0)
        hipLaunchKernelGGL(k_count_degree, dim3(grid_for((long long)c->flat_len, 256)), dim3(256), 0, st, cv, deg.p, N, flags.p);
    // exclusive scan over N+1 entries (last = total)
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, scan_bytes, deg.p, c->n2e_off.p, N + 1, st));
    HIP_TRY(c, c->n2e.alloc((size_t)c->flat_len + 1));
    if (c->flat_len > 0) {
        hipLaunchKernelGGL(k_fill_n2e, dim3(grid_for((long long)c->flat_len, 256)), dim3(256), 0, st, cv, c->n2e_off.p, cursor.p,
                           c->n2e.p, N);
        hipLaunchKernelGGL(k_sort_n2e, dim3(grid_for(N, 256, 1 << 30)), dim3(256), 0, st, c->n2e_off.p, c->n2e.p, N);
    }
    DevBuf<unsigned> heavy_bits;
    int nheavy = 0, heavy_grid = 0;
    const int heavy_words = (N + 31) / 32;
    // largest number of elements at a node (one more scan-sized reduction): decides between one neighbour pass and two
    HIP_TRY(c, hipcub::DeviceReduce::Max(tmp.p, red_bytes, deg.p, red.p, N + 1, st));
    unsigned h_red[4] = {0, 0, 0, 0};
    int h_flags[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(h_red, red.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(h_flags, flags.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (h_flags[0]) return c->fail(FH_BAD_ARGUMENT, "connectivity refers to a node index >= num_nodes");
    const unsigned long long max_cand = (unsigned long long)h_red[0] * (unsigned)c->ei.n;
    bool once = N > 0 && !c->ragged && max_cand <= 128ull;
    DevBuf<unsigned> rows64;
    if (once) {
        // every node has at most 64 (hexahedra) or 128 (tetrahedra: two per lane) candidates: one neighbour pass into scratch rows of 64
        // distinct neighbours, scan, compaction (pattern_kernels.hpp); a node with more than 64 distinct ones sends the build to the two passes
        if (rows64.alloc((size_t)N * 64) != hipSuccess) {   // no room for the scratch rows: the two passes need none
            (void)hipGetLastError();
            once = false;
        }
    }
    if (once) {
        const int g = std::min(N, 256 * 64);
        if (max_cand <= 64ull) hipLaunchKernelGGL(k_node_neighbors_once<1>, dim3(g), dim3(64), 0, st, c->conn.p, c->ei.n, c->n2e_off.p, c->n2e.p, N, cnt.p, rows64.p, flags.p + 1);
        else hipLaunchKernelGGL(k_node_neighbors_once<2>, dim3(g), dim3(64), 0, st, c->conn.p, c->ei.n, c->n2e_off.p, c->n2e.p, N, cnt.p, rows64.p, flags.p + 1);
        HIP_TRY(c, hipGetLastError());
        if (max_cand > 64ull) {
            int over = 0;
            HIP_TRY(c, hipMemcpyAsync(&over, flags.p + 1, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            if (over) {
                once = false;
                rows64.release();
                HIP_TRY(c, hipMemsetAsync(flags.p + 1, 0, sizeof(int), st));
            }
        }
    }
    if (!once && N > 0) {
        const int g = std::min(N, 256 * 64);
        hipLaunchKernelGGL(k_node_neighbors<false>, dim3(g), dim3(64), 0, st, cv, c->n2e_off.p, c->n2e.p, N, cnt.p, nullptr,
                           nullptr, flags.p + 1, heavy.p);
        // nodes with more candidates than the LDS sort takes (none on a finite element mesh): counted through a bitmap
        HIP_TRY(c, hipMemcpyAsync(&nheavy, flags.p + 1, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (nheavy > HEAVY_CAP) return c->fail(FH_UNSUPPORTED, "more than 4096 nodes with more than 4096 candidate neighbours each");
        if (nheavy > 0) {
            heavy_grid = std::min(nheavy, 32);
            HIP_TRY(c, heavy_bits.alloc((size_t)heavy_grid * heavy_words));
            hipLaunchKernelGGL(k_heavy_neighbors<false>, dim3(heavy_grid), dim3(256), 0, st, cv, c->n2e_off.p, c->n2e.p, heavy.p, nheavy, N,
                               heavy_bits.p, heavy_words, cnt.p, nullptr, nullptr);
            HIP_TRY(c, hipGetLastError());
        }
    }
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, scan_bytes, cnt.p, c->noff.p, N + 1, st));
    // the longest row and the number of entries: two numbers come back (the host copies of the offsets are made when the gather
    // partition or fh_pattern's output needs them: host_offsets)
    HIP_TRY(c, hipcub::DeviceReduce::Max(tmp.p, red_bytes, cnt.p, red.p + 1, N + 1, st));
    HIP_TRY(c, hipMemcpyAsync(h_red + 1, red.p + 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(h_red + 2, c->noff.p + N, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    c->h_noff.clear();
    c->h_n2e_off.clear();
    c->nnz_nodes = h_red[2];
    c->max_row = h_red[1];
    // (a sum beyond 2^32 wraps the 32-bit scan: bounded through the candidates, sum of deg * n = flat_len * n at most)
    if (c->ragged || (unsigned long long)c->flat_len * (unsigned long long)c->ei.n >= (1ull << 32) - 1) {
        // exact check on the host for the meshes near the limit
        std::vector<unsigned> hc((size_t)N + 1);
        HIP_TRY(c, hipMemcpy(hc.data(), cnt.p, sizeof(unsigned) * ((size_t)N + 1), hipMemcpyDeviceToHost));
        unsigned long long tot = 0;
        for (int i = 0; i < N; ++i) tot += hc[i];
        if (tot >= (1ull << 32) - 1) return c->fail(FH_UNSUPPORTED, "node-level nnz exceeds 2^32");
    }
    HIP_TRY(c, c->ncols.alloc((size_t)c->nnz_nodes + 1));
    if (once) {
        hipLaunchKernelGGL(k_compact_neighbors, dim3((N + 255) / 256), dim3(256), 0, st, c->noff.p, rows64.p, N, c->ncols.p);
        HIP_TRY(c, hipGetLastError());
    } else if (N > 0) {
        const int g = std::min(N, 256 * 64);
        hipLaunchKernelGGL(k_node_neighbors<true>, dim3(g), dim3(64), 0, st, cv, c->n2e_off.p, c->n2e.p, N, nullptr, c->noff.p,
                           c->ncols.p, flags.p + 1, heavy.p);
        if (nheavy > 0)
            hipLaunchKernelGGL(k_heavy_neighbors<true>, dim3(heavy_grid), dim3(256), 0, st, cv, c->n2e_off.p, c->n2e.p, heavy.p, nheavy, N,
                               heavy_bits.p, heavy_words, nullptr, c->noff.p, c->ncols.p);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipGetLastError());
    c->has_pattern = true;
    ++c->pattern_gen;
    c->has_partition = false; ++c->struct_gen; c->has_tp_pos = false;
    return build_compute_adjacency(c);
}

// ---------------------------------------------------------------------------------- kernel dispatch
int check_ready(fh_ctx* c, const char* who, bool need_pattern) {
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, std::string(who) + ": no finite element mesh set");
    if (c->op < 0) return c->fail(FH_INVALID_STATE, std::string(who) + ": no operator set");
    if (c->nq <= 0) return c->fail(FH_INVALID_STATE, std::string(who) + ": no quadrature table set");
    if (c->op == FH_TENSOR) {
        if (!c->tensor.p || c->tensor_nq != c->nq)
            return c->fail(FH_INVALID_STATE, std::string(who) + ": FH_TENSOR needs fh_set_operator_tensor with one tensor per point of the quadrature table in use");
    } else if (c->op != FH_LAPLACE && !c->has_params)
        return c->fail(FH_INVALID_STATE, std::string(who) + ": operator needs per-point parameters (mu, lambda)");
    if (need_pattern && !c->has_pattern) return c->fail(FH_INVALID_STATE, std::string(who) + ": call fh_pattern first");
    return FH_OK;
}

// the pre-scaled-gradient ("fast") form of every kernel but the pipelined gather needs ONE uniform parameter pair;
// with a compact table only the pipelined kernel knows per-element data (fh_ctx::elem_par)

void fill_common(fh_ctx* c, KArgs& a) {
    std::memset(&a, 0, sizeof a);
    a.verts = c->verts.p;
    a.conn = c->conn.p;
    a.num_elements = (long long)c->E;
    a.num_nodes = (int)c->N;
    a.nq = c->nq;
    a.qw = c->qw.p;
    a.gref = c->gref.p;
    a.gref_t = c->gref_t.p;
    a.vtx_pack = 0;
    a.ke_tri = 0;
    a.ggeom = c->ggeom.p;
    a.phiref = c->phiref.p;
    // (the all-affine instantiations of the element pass drop the mixed coefficients of the geometry map for the residual and the energy of EVERY
    // operator: only under the default tolerance, where that is below rounding -- a caller who loosened fh_set_affine_tolerance gets the
    // stiffness fast path its documentation promises and exact geometry everywhere else, consistent with the two-pass tangent)
    a.all_affine = (c->elem_kind == FH_HEX8 && c->has_aff && c->num_aff == c->E && c->E > 0 && c->affine_tol <= 0x1p-46 &&
                    !c->env("FENRIS_HIP_NO_AFFINE_PASS")) ? 1 : 0;
    a.qmono = (c->elem_kind == FH_HEX8 && c->qmono.p && !c->env("FENRIS_HIP_NO_MONOMIAL")) ? c->qmono.p : nullptr;
    a.qmom = (a.qmono && a.all_affine && c->qmom_ok && c->qmom.p && !c->has_rules && (c->op == FH_LAPLACE || (c->op == FH_LINEAR_ELASTIC && c->has_params)) &&
              c->env_int("FENRIS_HIP_NO_MOMENT_RESIDUAL", 0) == 0) ? c->qmom.p : nullptr;
    a.qparams = c->has_params ? c->qparams.p : nullptr;
    a.tensor = c->op == FH_TENSOR ? c->tensor.p : nullptr;
    a.nonsym = (c->op == FH_TENSOR && !c->tensor_sym) ? 1 : 0;
    a.rule_map = c->has_rules ? c->rule_map.p : nullptr;
    a.rparams = c->has_rules ? c->rparams.p : nullptr;
    a.u = c->has_u ? c->u.p : nullptr;
    a.fast = (generic_fast(c) && (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC)) ? 1 : 0;
    a.mu = c->uni_mu;
    a.lambda = c->uni_lambda;
    a.noff = c->noff.p;
    a.ncols = c->ncols.p;
    a.n2e_off = c->n2e_off.p;
    a.n2e = c->n2e.p;
    a.status = c->status.p + c->status_slot;
    a.ablate = c->env_int("FENRIS_HIP_ABLATE", 0);
    a.trace = nullptr;
    if (c->env("FENRIS_HIP_TRACE")) {
        if (!c->trace.p && c->trace.alloc(32) == hipSuccess) (void)hipMemset(c->trace.p, 0, 256);
        a.trace = c->trace.p;
    }
}

int reset_status(fh_ctx* c) {
    if (!c->status.p || c->status.n < 2) {
        HIP_TRY(c, c->status.alloc(2));
        const DevStatus z[2] = {{0, 0, ~0ull}, {0, 0, ~0ull}};
        HIP_TRY(c, hipMemcpyAsync(c->status.p, z, sizeof z, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // z is on the stack
    }
    static const DevStatus s{0, 0, ~0ull};
    HIP_TRY(c, hipMemcpyAsync(c->status.p + c->status_slot, &s, sizeof s, hipMemcpyHostToDevice, c->stream));
    return FH_OK;
}

int read_status(fh_ctx* c, uint64_t* failed) {
    if (c->defer_status) return FH_OK;   // an _async entry point: nothing is waited for here
    // slot 0: the context's own launches; slot 1: fh_assemble_matrix_rows_dev (reset by the next such call, or here once
    // its error has been reported)
    DevStatus s[2] = {};
    const bool two = c->status.n >= 2;
    HIP_TRY(c, hipMemcpyAsync(s, c->status.p, sizeof(DevStatus) * (two ? 2 : 1), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    for (int k = 0; k < (two ? 2 : 1); ++k)
        if (s[k].singular) {
            if (failed) *failed = s[k].failed_elem;
            if (k == 1) {
                static const DevStatus z{0, 0, ~0ull};
                HIP_TRY(c, hipMemcpyAsync(c->status.p + 1, &z, sizeof z, hipMemcpyHostToDevice, c->stream));
            }
            return c->fail(FH_SINGULAR_JACOBIAN, "Singular element Jacobian encountered");
        }
    return FH_OK;
}

// choose elements-per-block for the element-centric kernels so the LDS footprint stays <= target
static int upload_colors(fh_ctx* c, const std::vector<uint64_t>& offs, const std::vector<uint64_t>& labels) {
    // with an element mask only the active elements of each colour are launched
    std::vector<unsigned> l32;
    std::vector<uint64_t> o2(1, 0);
    l32.reserve(labels.size() + 1);
    for (size_t col = 0; col + 1 < offs.size(); ++col) {
        for (uint64_t k = offs[col]; k < offs[col + 1]; ++k)
            if (!c->has_mask || c->h_active[labels[k]]) l32.push_back((unsigned)labels[k]);
        o2.push_back(l32.size());
    }
    if (l32.empty()) l32.push_back(0);
    HIP_TRY(c, c->labels.alloc(l32.size()));
    HIP_TRY(c, hipMemcpy(c->labels.p, l32.data(), sizeof(unsigned) * l32.size(), hipMemcpyHostToDevice));
    c->color_offsets = o2;
    c->host_colors_offs = offs;
    c->host_colors_labels = labels;
    c->has_colors = true;
    return FH_OK;
}


// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int fh_abi_version(void) { return FH_ABI_VERSION; }

fh_ctx* fh_create(int device_id) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count) return nullptr;
    DevGuard dev_guard_(device_id);   // the calling thread's current device is left as it was
    fh_ctx* c = new fh_ctx();
    c->device = device_id;
    // the tuning / diagnostic switches, once (include/fenris_hip.h): nothing in the dispatch reads the environment later
    for (char** ev = environ; ev && *ev; ++ev) {
        if (std::strncmp(*ev, "FENRIS_HIP_", 11) != 0) continue;
        const char* eq = std::strchr(*ev, '=');
        if (eq) c->env_vars.emplace(std::string(*ev, (size_t)(eq - *ev)), std::string(eq + 1));
    }
    const DevStatus z[2] = {{0, 0, ~0ull}, {0, 0, ~0ull}};
    if (c->status.alloc(2) != hipSuccess || hipMemcpy(c->status.p, z, sizeof z, hipMemcpyHostToDevice) != hipSuccess) {
        delete c;
        return nullptr;
    }
    return c;
}

void fh_destroy(fh_ctx* c) {
    if (!c) return;
    DevGuard dev_guard_(c->device);
    if (c->trace.p) {  // FENRIS_HIP_TRACE: average cycles per wave and phase of the pipelined kernel
        unsigned long long h[32] = {0};
        (void)hipDeviceSynchronize();
        if (hipMemcpy(h, c->trace.p, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && (h[6] || h[27])) {   // (h[27]: the records wave of the fused affine kernel reports as "wave 3": landed-wait, arithmetic, DMA issue, barrier)
            // pipelined kernel: six phases of waves 0-3; affine kernel: "wave" = role (0 row wave, 1 loader, 2 store wave), phase 0 =
            // work between barriers, phase 2 = at the barrier
            static const char* names_pipe[6] = {"top", "phaseB+writeout(prev)", "barrier", "phaseC", "finalize+park", "end barrier"};
            // k_affine_ring: role 0 = row wave 0, 1 = loader wave, 2 = store wave
            static const char* names_ring[3][6] = {{"work", "wait: loader", "wait: ring space", "drain + publish", "-", "-"},
                                                   {"other", "wait: rows / store", "park + issue (vmcnt)", "headers, tables, publish", "-", "-"},
                                                   {"stream", "wait: rows", "drain + publish", "-", "-", "-"}};
            // k_hex8_rows: role 0 = row wave 0, 1 = row wave 3, 2 = loader wave, 3 = store wave; the two halves of a position and their barriers
            static const char* names_hex8[6] = {"first half (phase B | stream)", "barrier 1", "second half (phase C | loads)", "barrier 2", "-", "-"};
            const bool hex8l = h[30] == 0x48455838ull;
            const bool ringl = h[30] == 0x52494E47ull;
            for (int w = 0; w < 4; ++w) {
                const char* const* names = hex8l ? names_hex8 : (ringl && w < 3) ? names_ring[w] : names_pipe;
                const unsigned long long* r = h + 7 * w;
                if (!r[6]) continue;
                unsigned long long tot = 0;
                for (int k = 0; k < 6; ++k) tot += r[k];
                for (int k = 0; k < 6; ++k)
                    std::fprintf(stderr, "[fenris_hip trace] wave %d %-24s %12.0f cycles/wave  %5.1f %%\n", w, names[k],
                                 (double)r[k] / (double)r[6], 100.0 * (double)r[k] / (double)tot);
            }
        }
    }
    if (c->trace.p) {  // ... and of k_hex27_dense_mfma (hex27_mfma.hpp): cycles of wavefront 0 per phase and element
        unsigned long long h[32] = {0};
        if (hipMemcpy(h, c->trace.p, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[31]) {
            static const char* names_tiles[9] = {"P0 inputs", "P1 J, inverse", "P2 gradients", "P3 grad u", "P4 F, coefficients", "P5 F^-T g",
                                                 "round A + its stores", "round B", "stores of round B"};
            // (h[30] == 2: hex27_blocks.hpp, five phases)
            static const char* names_blocks[9] = {"P0 inputs + barrier", "P1 chain + barrier", "P2 a = M^T r + barrier", "matrix phase", "stores", "", "", "", ""};
            const char* const* names = h[30] == 2 ? names_blocks : names_tiles;
            const int nph = h[30] == 2 ? 5 : 9;
            unsigned long long tot = 0;
            for (int k = 0; k < nph; ++k) tot += h[16 + k];
            for (int k = 0; k < nph; ++k)
                std::fprintf(stderr, "[fenris_hip trace] hex27 %-24s %10.0f cycles/element  %5.1f %%\n", names[k],
                             (double)h[16 + k] / (double)h[31], 100.0 * (double)h[16 + k] / (double)tot);
        }
    }
    delete c->rows_stash;
    delete c;
}

const char* fh_last_error(const fh_ctx* c) { return c ? c->err.c_str() : "null context"; }
} // extern "C"
// accessors for group.hip (group_internal.hpp)
int fh_internal_fail(fh_ctx* c, int code, const std::string& msg) { return c ? c->fail(code, msg) : code; }
int fh_internal_device(const fh_ctx* c) { return c->device; }
hipStream_t fh_internal_stream(const fh_ctx* c) { return c->stream; }
bool fh_internal_pattern(const fh_ctx* c, const unsigned** noff, const unsigned** ncols, uint64_t* num_nodes, int* solution_dim) {
    if (!c->has_pattern) return false;
    *noff = c->noff.p; *ncols = c->ncols.p; *num_nodes = c->N; *solution_dim = c->S();
    return true;
}
unsigned long long fh_internal_pattern_gen(const fh_ctx* c) { return c->has_pattern ? c->pattern_gen : 0ull; }
bool fh_internal_num_nodes(const fh_ctx* c, uint64_t* num_nodes) {
    if (!c->has_mesh) return false;
    *num_nodes = c->N;
    return true;
}
bool fh_internal_sizes(const fh_ctx* c, uint64_t* num_nodes, int* solution_dim) {
    if (!c->has_mesh || (c->op < 0 && !c->ragged)) return false;
    *num_nodes = c->N; *solution_dim = c->S();
    return true;
}
extern "C" {
const char* fh_last_kernel_name(const fh_ctx* c) { return c ? c->last_kernel.c_str() : ""; }

// A tuning switch of THIS context (the FENRIS_HIP_* names, read by fh_create from the environment): set or, with value == NULL,
// removed.  Switches that select a launch variant take effect at the next call; those that shape tables need the tables rebuilt
// (fh_set_operator / fh_set_mesh).  For experiments that compare variants inside one context, on the same buffers -- the only
// comparison that resolves less than ~4 %.
int fh_set_option(fh_ctx* c, const char* name, const char* value) {
    if (!c || !name) return FH_BAD_ARGUMENT;
    if (value) c->env_vars[name] = value; else c->env_vars.erase(name);
    return FH_OK;
}

int fh_set_stream(fh_ctx* c, void* s) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    c->stream = reinterpret_cast<hipStream_t>(s);
    return FH_OK;
}
int fh_synchronize(fh_ctx* c) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FH_OK;
}

uint64_t fh_host_pool_trim(void) { return (uint64_t)HostPool::trim(); }
uint64_t fh_solution_dim(const fh_ctx* c) { return c ? (uint64_t)c->S() : 0; }
uint64_t fh_num_elements(const fh_ctx* c) { return c ? c->E : 0; }
uint64_t fh_num_nodes(const fh_ctx* c) { return c ? c->N : 0; }
uint64_t fh_num_rows(const fh_ctx* c) { return c ? (uint64_t)c->S() * c->N : 0; }
uint64_t fh_nnz(const fh_ctx* c) { return (c && c->has_pattern) ? (uint64_t)c->S() * c->S() * c->nnz_nodes : 0; }

// per-element affine flags from the current vertex coordinates (Hex8; affine_kernel.hpp).  The owner-computes partition
// depends on them: it is rebuilt when they change.
static int classify_affine(fh_ctx* c) {
    const bool had = c->has_aff;
    const uint64_t old_count = c->num_aff;
    c->has_aff = false;
    c->num_aff = 0;
    if (c->elem_kind != FH_HEX8 || c->E == 0 || !(c->affine_tol > 0.0)) {
        if (had) c->has_partition = false;
        ++c->struct_gen;   // always: the rows stash follows the classification through the generation counter
        return FH_OK;
    }
    DevBuf<unsigned char> flags;
    DevBuf<unsigned long long> cnt;
    HIP_TRY(c, flags.alloc((size_t)c->E));
    HIP_TRY(c, cnt.alloc(1));
    HIP_TRY(c, hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_classify_affine_hex8, dim3((unsigned)((c->E + 255) / 256)), dim3(256), 0, c->stream, c->verts.p, c->conn.p,
                       (long long)c->E, c->affine_tol, flags.p, cnt.p);
    HIP_TRY(c, hipGetLastError());
    unsigned long long h = 0;
    HIP_TRY(c, hipMemcpyAsync(&h, cnt.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // same flags as before (the usual case of fh_update_vertices on a moving mesh: none before, none now; or a rigid motion):
    // keep the partition.  Equal counts with different members are told apart by comparing the arrays.
    bool same = had && old_count == h && c->elem_aff.n >= c->E;
    if (same && h != 0 && h != c->E) {
        DevBuf<int> diff;
        HIP_TRY(c, diff.alloc(1));
        HIP_TRY(c, hipMemsetAsync(diff.p, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(k_bytes_differ, dim3((unsigned)((c->E + 255) / 256)), dim3(256), 0, c->stream, flags.p, c->elem_aff.p,
                           (long long)c->E, diff.p);
        int hd = 0;
        HIP_TRY(c, hipMemcpyAsync(&hd, diff.p, sizeof hd, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        same = hd == 0;
    }
    std::swap(c->elem_aff.p, flags.p);
    std::swap(c->elem_aff.n, flags.n);
    c->num_aff = h;
    c->has_aff = true;
    if (!same) { c->has_partition = false; ++c->struct_gen; c->aff_failed = false; }
    // The element records of the affine kernel are a long-lived buffer of E x 80 bytes that is read by every assembly: it is
    // reserved NOW, while the device memory of a fresh context is still unfragmented -- allocated at the first assembly it lands in
    // whatever the table builders' temporaries left behind, and the time of the headline kernel follows how its buffers happen to be
    // backed (two modes 8 % apart, profiles/r03_affine_experiments.txt).
    if (h > 0 && c->a_recs.n < (size_t)c->E * AFFINE_ROWS_GW_LE)
        HIP_TRY(c, c->a_recs.alloc((size_t)c->E * AFFINE_ROWS_GW_LE));
    return FH_OK;
}

static int set_mesh_common(fh_ctx* c, int elem_kind, uint64_t N, uint64_t E) {
    ElemInfo ei;
    if (!elem_info(elem_kind, ei)) return c->fail(FH_BAD_ARGUMENT, "fh_set_mesh: unknown element kind");
    if (N >= (1ull << 31)) return c->fail(FH_UNSUPPORTED, "fh_set_mesh: num_vertices must be < 2^31");
    if (E * (uint64_t)ei.n >= (1ull << 32)) return c->fail(FH_UNSUPPORTED, "fh_set_mesh: num_elements * n must be < 2^32");
    HIP_TRY(c, hipSetDevice(c->device));
    invalidate_pattern(c);
    c->has_mesh = false;
    c->ragged = false;
    c->elem_kind = elem_kind;
    c->ei = ei;
    c->N = N;
    c->E = E;
    c->flat_len = E * (uint64_t)ei.n;
    c->has_u = false;
    c->has_mask = false;
    c->has_aff = false;
    c->aff_failed = false;
    c->perm_failed = false;
    c->rows_try = 0;
    c->has_ghat = false;
    c->rs.active = false;  // rule-set tables and element masks are per-mesh state
    c->user_has_mask = false;
    c->user_mask.clear();
    c->row_lo = 0;   // the row range is per-mesh state
    c->row_hi = -1;
    c->nq = 0;  // reference gradient tables depend on the element kind
    HIP_TRY(c, c->verts.alloc((size_t)N * ei.d));
    HIP_TRY(c, c->conn.alloc((size_t)c->flat_len));
    ++c->topo_gen;   // a new connectivity: tables that depend on the topology alone (the element tiles) are rebuilt
    return FH_OK;
}

static int narrow_conn(fh_ctx* c, const unsigned long long* conn_dev) {
    DevBuf<int> bad;
    HIP_TRY(c, bad.alloc(1));
    HIP_TRY(c, hipMemsetAsync(bad.p, 0, sizeof(int), c->stream));
    if (c->flat_len)
        hipLaunchKernelGGL(k_narrow_connectivity, dim3(grid_for((long long)c->flat_len, 256)), dim3(256), 0, c->stream, conn_dev,
                           c->conn.p, (long long)c->flat_len, (int)c->N, bad.p);
    int h = 0;
    HIP_TRY(c, hipMemcpyAsync(&h, bad.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h) return c->fail(FH_BAD_ARGUMENT, "fh_set_mesh: connectivity refers to a vertex index >= num_vertices");
    return FH_OK;
}

int fh_set_mesh(fh_ctx* c, int elem_kind, const double* vertices, uint64_t N, const uint64_t* connectivity, uint64_t E) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if ((N && !vertices) || (E && !connectivity)) return c->fail(FH_BAD_ARGUMENT, "fh_set_mesh: null pointer");
    int rc = set_mesh_common(c, elem_kind, N, E);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->verts.p, vertices, sizeof(double) * N * c->ei.d, hipMemcpyHostToDevice, c->stream));
    DevBuf<unsigned long long> tmp;
    HIP_TRY(c, tmp.alloc((size_t)c->flat_len));
    HIP_TRY(c, hipMemcpyAsync(tmp.p, connectivity, sizeof(uint64_t) * c->flat_len, hipMemcpyHostToDevice, c->stream));
    rc = narrow_conn(c, tmp.p);
    if (rc) return rc;
    // host copy for the (host-side, reference-identical) greedy colouring
    c->h_nodes.assign(connectivity, connectivity + c->flat_len);
    c->h_eoff.clear();
    c->has_host_conn = true;
    c->has_mesh = true;
    return classify_affine(c);
}

int fh_set_mesh_dev(fh_ctx* c, int elem_kind, const double* vertices_dev, uint64_t N, const uint64_t* conn_dev, uint64_t E) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if ((N && !vertices_dev) || (E && !conn_dev)) return c->fail(FH_BAD_ARGUMENT, "fh_set_mesh_dev: null pointer");
    int rc = set_mesh_common(c, elem_kind, N, E);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->verts.p, vertices_dev, sizeof(double) * N * c->ei.d, hipMemcpyDeviceToDevice, c->stream));
    rc = narrow_conn(c, reinterpret_cast<const unsigned long long*>(conn_dev));
    if (rc) return rc;
    c->h_nodes.clear();
    c->has_host_conn = false;
    c->has_mesh = true;
    return classify_affine(c);
}

int fh_update_vertices(fh_ctx* c, const double* vertices) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, "fh_update_vertices: no mesh");
    if (!vertices) return c->fail(FH_BAD_ARGUMENT, "fh_update_vertices: null pointer");
    HIP_TRY(c, hipMemcpyAsync(c->verts.p, vertices, sizeof(double) * c->N * c->ei.d, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return classify_affine(c);
}

int fh_set_connectivity_ragged(fh_ctx* c, uint64_t sdim, uint64_t N, const uint64_t* eoff, const uint64_t* nodes, uint64_t E) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!eoff || sdim == 0) return c->fail(FH_BAD_ARGUMENT, "fh_set_connectivity_ragged: bad argument");
    const uint64_t total = eoff[E];
    if (total && !nodes) return c->fail(FH_BAD_ARGUMENT, "fh_set_connectivity_ragged: null node list");
    if (N >= (1ull << 31) || total >= (1ull << 32)) return c->fail(FH_UNSUPPORTED, "connectivity too large");
    HIP_TRY(c, hipSetDevice(c->device));
    invalidate_pattern(c);
    c->has_mesh = false;
    c->ragged = true;
    c->has_aff = false;
    c->row_lo = 0;
    c->row_hi = -1;
    c->elem_kind = -1;
    c->ei = ElemInfo{0, 0, 0, -1};
    c->N = N;
    c->E = E;
    c->flat_len = total;
    c->sdim_ragged = sdim;
    std::vector<int> h_nodes(total ? total : 1, 0);
    std::vector<unsigned> h_off(E + 1), h_k2e(total ? total : 1, 0);
    for (uint64_t e = 0; e <= E; ++e) {
        if (e && eoff[e] < eoff[e - 1]) return c->fail(FH_BAD_ARGUMENT, "element offsets must be non-decreasing");
        h_off[e] = (unsigned)eoff[e];
    }
    for (uint64_t e = 0; e < E; ++e)
        for (uint64_t k = eoff[e]; k < eoff[e + 1]; ++k) {
            if (nodes[k] >= N) return c->fail(FH_BAD_ARGUMENT, "connectivity refers to a node index >= num_nodes");
            h_nodes[k] = (int)nodes[k];
            h_k2e[k] = (unsigned)e;
        }
    HIP_TRY(c, c->conn.alloc(h_nodes.size()));
    HIP_TRY(c, c->eoff.alloc(h_off.size()));
    HIP_TRY(c, c->k2e.alloc(h_k2e.size()));
    HIP_TRY(c, hipMemcpy(c->conn.p, h_nodes.data(), sizeof(int) * h_nodes.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->eoff.p, h_off.data(), sizeof(unsigned) * h_off.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->k2e.p, h_k2e.data(), sizeof(unsigned) * h_k2e.size(), hipMemcpyHostToDevice));
    c->h_eoff.assign(eoff, eoff + E + 1);
    c->h_nodes.assign(nodes, nodes + total);
    c->has_host_conn = true;
    c->has_mesh = true;
    return FH_OK;
}

int fh_set_active_elements(fh_ctx* c, const uint8_t* mask) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, "fh_set_active_elements: set the mesh first");
    c->user_has_mask = mask != nullptr;
    if (mask) c->user_mask.assign(mask, mask + c->E); else c->user_mask.clear();
    return apply_mask(c, mask);
}
extern "C++" int apply_mask(fh_ctx* c, const uint8_t* mask) {
    c->has_partition = false; ++c->struct_gen; c->has_tp_pos = false;
    if (!mask) {
        c->has_mask = false;
        if (c->has_colors) return upload_colors(c, c->host_colors_offs, c->host_colors_labels);
        return FH_OK;
    }
    c->h_active.assign(mask, mask + c->E);
    std::vector<unsigned> list;
    list.reserve(c->E);
    for (uint64_t e = 0; e < c->E; ++e) {
        c->h_active[e] = mask[e] ? 1 : 0;
        if (mask[e]) list.push_back((unsigned)e);
    }
    c->num_active = list.size();
    if (list.empty()) list.push_back(0);
    HIP_TRY(c, c->active.alloc((size_t)c->E + 1));
    HIP_TRY(c, c->active_list.alloc(list.size()));
    HIP_TRY(c, hipMemcpy(c->active.p, c->h_active.data(), c->E, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->active_list.p, list.data(), sizeof(unsigned) * list.size(), hipMemcpyHostToDevice));
    c->has_mask = true;
    if (c->has_colors) {
        int rc = upload_colors(c, c->host_colors_offs, c->host_colors_labels);
        if (rc) return rc;
    }
    return build_compute_adjacency(c);
}

int fh_set_row_range(fh_ctx* c, uint64_t node_begin, uint64_t node_end) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, "fh_set_row_range: set the mesh first");
    if (node_begin > node_end || node_end > c->N) return c->fail(FH_BAD_ARGUMENT, "fh_set_row_range: bad node range");
    if (node_begin == 0 && node_end == c->N) { c->row_lo = 0; c->row_hi = -1; }
    else { c->row_lo = (long long)node_begin; c->row_hi = (long long)node_end; }
    c->has_partition = false; ++c->struct_gen; c->has_tp_pos = false;
    return FH_OK;
}

int fh_set_affine_tolerance(fh_ctx* c, double rel_tol) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!(rel_tol >= 0.0) || rel_tol > 1e-6) return c->fail(FH_BAD_ARGUMENT, "fh_set_affine_tolerance: tolerance must be in [0, 1e-6]");
    if (rel_tol == c->affine_tol) return FH_OK;
    c->affine_tol = rel_tol;
    if (c->has_mesh && !c->ragged) return classify_affine(c);
    return FH_OK;
}

int fh_affine_stats(const fh_ctx* c, uint64_t* affine_elements, uint64_t* affine_blocks, uint64_t* general_blocks) {
    if (!c) return FH_BAD_ARGUMENT;
    if (affine_elements) *affine_elements = c->has_aff ? c->num_aff : 0;
    if (affine_blocks) *affine_blocks = c->has_partition ? (uint64_t)c->a_npos : 0;
    if (general_blocks) *general_blocks = c->has_partition ? (uint64_t)c->npos_gen : 0;
    return FH_OK;
}

int fh_set_operator(fh_ctx* c, int op_kind) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (op_kind < FH_LAPLACE || op_kind > FH_TENSOR) return c->fail(FH_BAD_ARGUMENT, "fh_set_operator: unknown operator");
    if (c->ragged) return c->fail(FH_INVALID_STATE, "fh_set_operator: context holds a ragged connectivity");
    const int old_s = c->S(), old_op = c->op;
    c->op = op_kind;
    if (c->S() != old_s) { c->has_u = false; c->has_tp_pos = false; }
    // the owner-computes partition (LDS budgets, kernel classes, slot parameters) is built for one operator
    if (op_kind != old_op) { c->has_partition = false; ++c->struct_gen; c->has_slotpar = false; c->perm_failed = false; c->rows_try = 0; }
    return FH_OK;
}

int fh_set_operator_tensor(fh_ctx* c, const double* tensors, uint32_t nq, int symmetric) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, "fh_set_operator_tensor: set the mesh first");
    if (!tensors || nq == 0) return c->fail(FH_BAD_ARGUMENT, "fh_set_operator_tensor: bad argument");
    if (c->nq > 0 && (int)nq != c->nq) return c->fail(FH_BAD_ARGUMENT, "fh_set_operator_tensor: one tensor per point of the quadrature table in use");
    const size_t d = (size_t)c->ei.d, len = (size_t)nq * d * d * d * d;
    HIP_TRY(c, c->tensor.alloc(len));
    HIP_TRY(c, hipMemcpy(c->tensor.p, tensors, sizeof(double) * len, hipMemcpyHostToDevice));
    c->tensor_nq = (int)nq;
    c->tensor_sym = symmetric != 0;
    return FH_OK;
}

int fh_set_quadrature_uniform(fh_ctx* c, const double* w, const double* pts, uint32_t nq, const double* params) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->rs_staging) c->rs.active = false;
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, "fh_set_quadrature_uniform: set the mesh first");
    if (!w || !pts || nq == 0) return c->fail(FH_BAD_ARGUMENT, "fh_set_quadrature_uniform: bad argument");
    const ElemInfo& ei = c->ei;
    std::vector<double> gref((size_t)nq * ei.n * ei.d), ggeom((size_t)nq * ei.ng * ei.d), phiref((size_t)nq * ei.n);
    std::vector<double> phigeom((size_t)nq * ei.ng);
    for (uint32_t q = 0; q < nq; ++q) ref_basis(ei.geom_kind, pts + (size_t)q * ei.d, phigeom.data() + (size_t)q * ei.ng);
    HIP_TRY(c, c->phigeom.alloc(phigeom.size()));
    HIP_TRY(c, hipMemcpy(c->phigeom.p, phigeom.data(), sizeof(double) * phigeom.size(), hipMemcpyHostToDevice));
    for (uint32_t q = 0; q < nq; ++q) {
        ref_gradients(c->elem_kind, pts + (size_t)q * ei.d, gref.data() + (size_t)q * ei.n * ei.d);
        ref_gradients(ei.geom_kind, pts + (size_t)q * ei.d, ggeom.data() + (size_t)q * ei.ng * ei.d);
        ref_basis(c->elem_kind, pts + (size_t)q * ei.d, phiref.data() + (size_t)q * ei.n);
    }
    HIP_TRY(c, c->phiref.alloc(phiref.size()));
    HIP_TRY(c, hipMemcpy(c->phiref.p, phiref.data(), sizeof(double) * phiref.size(), hipMemcpyHostToDevice));
    if (c->elem_kind == FH_HEX8) {   // the monomial form of the element pass (element_pass.hpp)
        std::vector<double> qm((size_t)nq * 8, 0.0);
        for (uint32_t q = 0; q < nq; ++q) {
            const double xi = pts[3 * q], eta = pts[3 * q + 1], zeta = pts[3 * q + 2];
            double* m = qm.data() + (size_t)q * 8;
            m[0] = xi; m[1] = eta; m[2] = zeta; m[3] = eta * zeta; m[4] = xi * zeta; m[5] = xi * eta;
        }
        HIP_TRY(c, c->qmono.alloc(qm.size()));
        HIP_TRY(c, hipMemcpy(c->qmono.p, qm.data(), sizeof(double) * qm.size(), hipMemcpyHostToDevice));
        // moments of the rule for the quadrature-free residual of affine elements (element_pass.hpp, AFFM = 2): usable when every moment
        // sum_q w_q xi^a eta^b zeta^c (a, b, c <= 2) with an odd power vanishes -- the tensor Gauss rules -- and, for operators with
        // parameters, when all points carry the same pair
        double mom[3][3][3];
        double scale = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                for (int d = 0; d < 3; ++d) {
                    double m = 0.0, ma = 0.0;
                    for (uint32_t q = 0; q < nq; ++q) {
                        const double t = w[q] * std::pow(pts[3 * q], a) * std::pow(pts[3 * q + 1], b) * std::pow(pts[3 * q + 2], d);
                        m += t;
                        ma += std::fabs(t);
                    }
                    mom[a][b][d] = m;
                    scale = std::max(scale, ma);
                }
        bool ok = scale > 0.0;
        for (int a = 0; a < 3 && ok; ++a)
            for (int b = 0; b < 3 && ok; ++b)
                for (int d = 0; d < 3 && ok; ++d)
                    if (((a | b | d) & 1) && std::fabs(mom[a][b][d]) > 1e-14 * scale) ok = false;
        if (params)
            for (uint32_t q = 1; q < nq && ok; ++q)
                if (params[2 * q] != params[0] || params[2 * q + 1] != params[1]) ok = false;
        c->qmom_ok = ok;
        const double qm8[8] = {mom[0][0][0], mom[2][0][0], mom[0][2][0], mom[0][0][2], mom[0][2][2], mom[2][0][2], mom[2][2][0], 0.0};
        HIP_TRY(c, c->qmom.alloc(8));
        HIP_TRY(c, hipMemcpy(c->qmom.p, qm8, sizeof qm8, hipMemcpyHostToDevice));
    } else {
        c->qmono.release();
        c->qmom_ok = false;
    }
    HIP_TRY(c, c->qw.alloc(nq + 1));  // [nq]: sum of the weights (collapsed rule of the affine simplices, see dispatch)
    HIP_TRY(c, c->gref.alloc(gref.size()));
    HIP_TRY(c, c->ggeom.alloc(ggeom.size()));
    HIP_TRY(c, c->qparams.alloc(2 * (size_t)nq));
    HIP_TRY(c, hipMemcpy(c->qw.p, w, sizeof(double) * nq, hipMemcpyHostToDevice));
    {
        double wsum = 0.0;
        for (uint32_t q = 0; q < nq; ++q) wsum += w[q];
        HIP_TRY(c, hipMemcpy(c->qw.p + nq, &wsum, sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(c, hipMemcpy(c->gref.p, gref.data(), sizeof(double) * gref.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->ggeom.p, ggeom.data(), sizeof(double) * ggeom.size(), hipMemcpyHostToDevice));
    if (c->elem_kind == FH_HEX27) {   // node-major copy for the per-point chain of the matrix-core first pass (hex27_blocks.hpp)
        std::vector<double> gt(gref.size());
        for (uint32_t q = 0; q < nq; ++q)
            for (int n = 0; n < ei.n; ++n)
                for (int d = 0; d < ei.d; ++d) gt[((size_t)n * nq + q) * ei.d + d] = gref[((size_t)q * ei.n + n) * ei.d + d];
        HIP_TRY(c, c->gref_t.alloc(gt.size()));
        HIP_TRY(c, hipMemcpy(c->gref_t.p, gt.data(), sizeof(double) * gt.size(), hipMemcpyHostToDevice));
        // the same two tables with the nodes in lexicographic order of their reference positions (engine_internal.hpp: hex27_perm): node n is
        // the one whose basis function is 1 at lattice point (ix, iy, iz) of {-1, 0, 1}^3; its place is ix + 3 iy + 9 iz
        bool found_all = true;
        for (int l = 0; l < 27; ++l) {
            const double xi[3] = {(double)(l % 3) - 1.0, (double)((l / 3) % 3) - 1.0, (double)(l / 9) - 1.0};
            double phi[27];
            ref_basis(FH_HEX27, xi, phi);
            int who = -1;
            for (int n = 0; n < 27; ++n) if (std::fabs(phi[n] - 1.0) < 1e-9) who = who < 0 ? n : 27;
            if (who < 0 || who >= 27) { found_all = false; break; }
            c->hex27_perm[who] = l;
        }
        c->has_hex27_perm = found_all;
        if (found_all) {
            std::vector<double> gl(gref.size()), gtl(gref.size());
            for (uint32_t q = 0; q < nq; ++q)
                for (int n = 0; n < 27; ++n)
                    for (int d = 0; d < 3; ++d) {
                        const double v = gref[((size_t)q * 27 + n) * 3 + d];
                        gl[((size_t)q * 27 + c->hex27_perm[n]) * 3 + d] = v;
                        gtl[((size_t)c->hex27_perm[n] * nq + q) * 3 + d] = v;
                    }
            HIP_TRY(c, c->gref_lex.alloc(gl.size()));
            HIP_TRY(c, c->gref_t_lex.alloc(gtl.size()));
            HIP_TRY(c, hipMemcpy(c->gref_lex.p, gl.data(), sizeof(double) * gl.size(), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->gref_t_lex.p, gtl.data(), sizeof(double) * gtl.size(), hipMemcpyHostToDevice));
            c->hex27_vtx_pack = 0;
            for (int g = 0; g < 8; ++g) c->hex27_vtx_pack |= (unsigned long long)c->hex27_perm[g] << (5 * g);
        }
    } else {
        c->gref_t.release();
        c->has_hex27_perm = false;
    }
    c->has_ghat = false;
    if (c->elem_kind == FH_HEX8) {
        // reference blocks of the affine-element kernel: Ghat_ab[c][d] = sum_q w_q ghat_a(xi_q)[c] ghat_b(xi_q)[d], summed in
        // table order; Ghat_ba is the exact transpose of Ghat_ab (the factors of each product commute)
        std::vector<double> gh((size_t)64 * (AFFINE_GW_LE + 2 * AFFINE_GW_LAP), 0.0);
        double* le = gh.data();
        double* lap = gh.data() + 64 * AFFINE_GW_LE;
        // third table (mass matrix of the affine elements, op FH_MASS_SCALAR): sum_q w_q rho_q phi_a phi_b in the first place of a Laplace-shaped
        // block, rho = the first parameter of a point (mass.rs:131-286)
        double* mass = gh.data() + 64 * (AFFINE_GW_LE + AFFINE_GW_LAP);
        if (params)
            for (int a = 0; a < 8; ++a)
                for (int b = 0; b < 8; ++b) {
                    double m = 0.0;
                    for (uint32_t q = 0; q < nq; ++q) m += (w[q] * params[2 * q]) * (phiref[(size_t)q * 8 + a] * phiref[(size_t)q * 8 + b]);
                    mass[(a * 8 + b) * AFFINE_GW_LAP] = m;
                }
        for (int a = 0; a < 8; ++a)
            for (int b = 0; b < 8; ++b) {
                double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                for (uint32_t q = 0; q < nq; ++q) {
                    const double* ga = gref.data() + ((size_t)q * 8 + a) * 3;
                    const double* gb = gref.data() + ((size_t)q * 8 + b) * 3;
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) G[i][j] += w[q] * (ga[i] * gb[j]);
                }
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) le[(a * 8 + b) * AFFINE_GW_LE + 3 * i + j] = G[i][j];
                double* l6 = lap + (a * 8 + b) * AFFINE_GW_LAP;
                l6[0] = G[0][0]; l6[1] = G[0][1] + G[1][0]; l6[2] = G[0][2] + G[2][0];
                l6[3] = G[1][1]; l6[4] = G[1][2] + G[2][1]; l6[5] = G[2][2];
            }
        HIP_TRY(c, c->ghat.alloc(gh.size()));
        HIP_TRY(c, hipMemcpy(c->ghat.p, gh.data(), sizeof(double) * gh.size(), hipMemcpyHostToDevice));
        c->has_ghat = true;
    }
    c->has_params = params != nullptr;
    if (params) HIP_TRY(c, hipMemcpy(c->qparams.p, params, sizeof(double) * 2 * nq, hipMemcpyHostToDevice));
    c->h_points.assign(pts, pts + (size_t)nq * ei.d);
    c->nq = (int)nq;
    c->fast_ok = true;
    for (uint32_t q = 0; q < nq; ++q) c->fast_ok = c->fast_ok && (w[q] >= 0.0);
    c->uni_mu = c->uni_lambda = 0.0;
    if (params) {
        c->uni_mu = params[0];
        c->uni_lambda = params[1];
        for (uint32_t q = 1; q < nq; ++q) c->fast_ok = c->fast_ok && params[2 * q] == params[0] && params[2 * q + 1] == params[1];
    }
    if (c->env("FENRIS_HIP_NO_FAST")) c->fast_ok = false;
    c->has_rules = false;
    c->elem_par = false;
    c->has_partition = false; ++c->struct_gen; c->has_tp_pos = false;
    return FH_OK;
}

// the per-point Parameters of the reference as the caller stores them: `stride` bytes from one point's record to the next,
// `kind` says what the first doubles of a record are
int fh_set_quadrature_uniform_data(fh_ctx* c, const double* w, const double* pts, uint32_t nq, const void* data, uint32_t stride,
                                   int kind) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (kind == FH_DATA_NONE || !data) return fh_set_quadrature_uniform(c, w, pts, nq, nullptr);
    const uint32_t need = (kind == FH_DATA_LAME) ? 16u : (kind == FH_DATA_DENSITY ? 8u : 0u);
    if (!need || stride < need || stride % 8u) return c->fail(FH_BAD_ARGUMENT, "fh_set_quadrature_uniform_data: bad kind or stride");
    std::vector<double> pairs((size_t)nq * 2, 0.0);
    for (uint32_t q = 0; q < nq; ++q) {
        const double* rec = reinterpret_cast<const double*>(static_cast<const char*>(data) + (size_t)q * stride);
        pairs[2 * q] = rec[0];
        if (kind == FH_DATA_LAME) pairs[2 * q + 1] = rec[1];
    }
    return fh_set_quadrature_uniform(c, w, pts, nq, pairs.data());
}

int fh_set_quadrature_compact(fh_ctx* c, const double* w, const double* pts, uint32_t nq, uint64_t num_rules,
                              const double* rule_params, const uint64_t* elem_to_rule) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!rule_params || !elem_to_rule || num_rules == 0) return c->fail(FH_BAD_ARGUMENT, "fh_set_quadrature_compact: bad argument");
    int rc = fh_set_quadrature_uniform(c, w, pts, nq, rule_params);  // tables; rule 0 stands in for the uniform data
    if (rc) return rc;
    std::vector<unsigned> map((size_t)c->E + 1, 0u);
    for (uint64_t e = 0; e < c->E; ++e) {
        // "Each rule index must correspond to a provided quadrature rule" (quadrature_table.rs:366-372 panics)
        if (elem_to_rule[e] >= num_rules) return c->fail(FH_BAD_ARGUMENT, "fh_set_quadrature_compact: rule index out of bounds");
        map[e] = (unsigned)elem_to_rule[e];
    }
    HIP_TRY(c, c->rule_map.alloc(map.size()));
    HIP_TRY(c, c->rparams.alloc((size_t)num_rules * nq * 2));
    HIP_TRY(c, hipMemcpy(c->rule_map.p, map.data(), sizeof(unsigned) * map.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->rparams.p, rule_params, sizeof(double) * (size_t)num_rules * nq * 2, hipMemcpyHostToDevice));
    c->has_rules = true;
    // rules that are constant over their points (piecewise-constant material) keep the pre-scaled-gradient form: the
    // pipelined kernel reads (mu, lambda) per element; anything else takes the per-point-coefficient kernels
    bool weights_ok = true, rules_const = true;
    for (uint32_t q = 0; q < nq; ++q) weights_ok = weights_ok && (w[q] >= 0.0);
    for (uint64_t r = 0; r < num_rules && rules_const; ++r)
        for (uint32_t q = 1; q < nq; ++q)
            rules_const = rules_const && rule_params[(r * nq + q) * 2] == rule_params[r * nq * 2] &&
                          rule_params[(r * nq + q) * 2 + 1] == rule_params[r * nq * 2 + 1];
    c->elem_par = weights_ok && rules_const && !c->env("FENRIS_HIP_NO_FAST") && !c->env("FENRIS_HIP_NO_ELEM_PAR");
    c->fast_ok = c->elem_par;
    c->has_slotpar = false;
    c->has_partition = false; ++c->struct_gen; c->has_tp_pos = false;
    return FH_OK;
}

// ---- rule-set tables (quadrature_table.rs:57-210 GeneralQuadratureTable, :300-439 CompactQuadratureTable)
extern "C++" int rs_stage(fh_ctx* c, int g) {
    auto& rs = c->rs;
    const auto& G = rs.groups[(size_t)g];
    const int d = c->ei.d;
    const uint32_t r0 = G[0];
    const uint32_t nq = (uint32_t)(rs.offs[r0 + 1] - rs.offs[r0]);
    const double* w = rs.w.data() + rs.offs[r0];
    const double* p = rs.pts.data() + rs.offs[r0] * (size_t)d;
    int rc;
    c->rs_staging = true;
    if (rs.par.empty()) {
        rc = fh_set_quadrature_uniform(c, w, p, nq, nullptr);
    } else if (G.size() == 1) {
        rc = fh_set_quadrature_uniform(c, w, p, nq, rs.par.data() + 2 * rs.offs[r0]);
    } else {  // rules that share points and weights and differ in their data: the compact device table
        std::vector<double> rp(G.size() * (size_t)nq * 2);
        for (size_t k = 0; k < G.size(); ++k)
            std::memcpy(rp.data() + k * nq * 2, rs.par.data() + 2 * rs.offs[G[k]], sizeof(double) * nq * 2);
        std::vector<uint64_t> local((size_t)c->E, 0);
        for (uint64_t el = 0; el < c->E; ++el)
            if (rs.rule_group[rs.e2r[el]] == g) local[el] = (uint64_t)rs.rule_local[rs.e2r[el]];
        rc = fh_set_quadrature_compact(c, w, p, nq, G.size(), rp.data(), local.data());
    }
    c->rs_staging = false;
    if (rc) return rc;
    std::vector<uint8_t> m((size_t)c->E);
    for (uint64_t el = 0; el < c->E; ++el)
        m[el] = (rs.rule_group[rs.e2r[el]] == g && (!c->user_has_mask || c->user_mask[el])) ? 1 : 0;
    rc = apply_mask(c, m.data());
    rs.staged = g;
    return rc;
}


int fh_set_quadrature_rules(fh_ctx* c, uint64_t num_rules, const uint64_t* rule_offsets, const double* weights, const double* points,
                            const double* params, const uint64_t* elem_to_rule) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_mesh || c->ragged || c->op < 0) return c->fail(FH_INVALID_STATE, "fh_set_quadrature_rules: set mesh and operator first");
    if (!rule_offsets || !weights || !points || num_rules == 0) return c->fail(FH_BAD_ARGUMENT, "fh_set_quadrature_rules: bad argument");
    if (!elem_to_rule && num_rules != c->E)
        return c->fail(FH_BAD_ARGUMENT, "fh_set_quadrature_rules: without a map there must be one rule per element");
    const size_t d = (size_t)c->ei.d;
    auto& rs = c->rs;
    rs.active = false;
    rs.offs.assign(rule_offsets, rule_offsets + num_rules + 1);
    for (uint64_t r = 0; r < num_rules; ++r)
        if (rs.offs[r + 1] <= rs.offs[r] || rs.offs[r + 1] - rs.offs[r] > 0xffffffffull)
            return c->fail(FH_BAD_ARGUMENT, "fh_set_quadrature_rules: every rule needs at least one point");
    const size_t total = (size_t)rs.offs[num_rules];
    rs.w.assign(weights, weights + total);
    rs.pts.assign(points, points + total * d);
    if (params) rs.par.assign(params, params + total * 2); else rs.par.clear();
    rs.e2r.resize((size_t)c->E);
    for (uint64_t el = 0; el < c->E; ++el) {
        const uint64_t r = elem_to_rule ? elem_to_rule[el] : el;
        // "Each rule index must correspond to a provided quadrature rule" (quadrature_table.rs:366-372 panics)
        if (r >= num_rules) return c->fail(FH_BAD_ARGUMENT, "fh_set_quadrature_rules: rule index out of bounds");
        rs.e2r[el] = (uint32_t)r;
    }
    // groups: rules with bitwise identical points and weights
    rs.rule_group.assign((size_t)num_rules, -1);
    rs.rule_local.assign((size_t)num_rules, 0);
    rs.groups.clear();
    std::unordered_map<std::string, int> seen;
    for (uint64_t r = 0; r < num_rules; ++r) {
        const size_t nq = (size_t)(rs.offs[r + 1] - rs.offs[r]);
        std::string key(reinterpret_cast<const char*>(rs.w.data() + rs.offs[r]), sizeof(double) * nq);
        key.append(reinterpret_cast<const char*>(rs.pts.data() + rs.offs[r] * d), sizeof(double) * nq * d);
        auto it = seen.find(key);
        if (it == seen.end()) {
            it = seen.emplace(std::move(key), (int)rs.groups.size()).first;
            rs.groups.emplace_back();
        }
        rs.rule_group[r] = it->second;
        rs.rule_local[r] = (int)rs.groups[(size_t)it->second].size();
        rs.groups[(size_t)it->second].push_back((uint32_t)r);
    }
    rs.active = true;
    // the first group stays staged (element-level queries see a valid table); the assemblers restage as they walk
    int rc = rs_stage(c, 0);
    const int rc2 = apply_mask(c, c->user_has_mask ? c->user_mask.data() : nullptr);
    if (rc || rc2) { rs.active = false; return rc ? rc : rc2; }
    return FH_OK;
}

int fh_quadrature_rule_groups(const fh_ctx* c, uint64_t* num_groups) {
    if (!c || !num_groups) return FH_BAD_ARGUMENT;
    *num_groups = c->rs.active ? c->rs.groups.size() : 0;
    return FH_OK;
}

static int set_u_common(fh_ctx* c, const double* u, hipMemcpyKind kind) {
    if (!c->has_mesh || c->ragged || c->op < 0) return c->fail(FH_INVALID_STATE, "fh_set_u: set mesh and operator first");
    if (!u) { c->has_u = false; return FH_OK; }
    const size_t len = (size_t)c->S() * c->N;
    if (c->u.n < len) HIP_TRY(c, c->u.alloc(len));
    HIP_TRY(c, hipMemcpyAsync(c->u.p, u, sizeof(double) * len, kind, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->has_u = true;
    return FH_OK;
}
int fh_set_u(fh_ctx* c, const double* u) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    return set_u_common(c, u, hipMemcpyHostToDevice);
}
int fh_set_u_dev(fh_ctx* c, const double* u) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    return set_u_common(c, u, hipMemcpyDeviceToDevice);
}

// ---- pattern
int fh_pattern_dev(fh_ctx* c, uint64_t* row_offsets_dev, uint64_t* col_indices_dev) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_pattern) return c->fail(FH_INVALID_STATE, "fh_pattern_dev: call fh_pattern first");
    const int S = c->S(), N = (int)c->N;
    if (row_offsets_dev)
        hipLaunchKernelGGL(k_expand_row_offsets, dim3(grid_for((long long)N * S + 1, 256, 1 << 30)), dim3(256), 0, c->stream,
                           c->noff.p, N, S, reinterpret_cast<unsigned long long*>(row_offsets_dev));
    if (col_indices_dev && c->nnz_nodes)
        hipLaunchKernelGGL(k_expand_col_indices, dim3(grid_for((long long)c->nnz_nodes, 256)), dim3(256), 0, c->stream, c->noff.p,
                           c->ncols.p, N, S, reinterpret_cast<unsigned long long*>(col_indices_dev));
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

int fh_pattern(fh_ctx* c, uint64_t* row_offsets, uint64_t* nnz_out) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = build_pattern(c);
    if (rc) return rc;
    const uint64_t S = (uint64_t)c->S();
    if (nnz_out) *nnz_out = S * S * c->nnz_nodes;
    if (row_offsets) {  // cheap on the host from the node-level offsets
        rc = host_offsets(c);
        if (rc) return rc;
        uint64_t r = 0;
        for (uint64_t i = 0; i < c->N; ++i) {
            const uint64_t cnt = c->h_noff[i + 1] - c->h_noff[i];
            for (uint64_t t = 0; t < S; ++t) row_offsets[r++] = S * S * c->h_noff[i] + t * S * cnt;
        }
        row_offsets[r] = S * S * c->nnz_nodes;
    }
    return FH_OK;
}

int fh_pattern_cols(fh_ctx* c, uint64_t* col_indices) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_pattern) return c->fail(FH_INVALID_STATE, "fh_pattern_cols: call fh_pattern first");
    const uint64_t nnz = fh_nnz(c);
    if (nnz == 0) return FH_OK;
    if (!col_indices) return c->fail(FH_BAD_ARGUMENT, "fh_pattern_cols: null pointer");
    DevBuf<unsigned long long> tmp;
    HIP_TRY(c, tmp.alloc((size_t)nnz));
    int rc = fh_pattern_dev(c, nullptr, reinterpret_cast<uint64_t*>(tmp.p));
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(col_indices, tmp.p, sizeof(uint64_t) * nnz, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FH_OK;
}

// ---- colouring
int fh_color(fh_ctx* c, uint64_t* num_colors, uint64_t* color_offsets, uint64_t* labels) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_mesh) return c->fail(FH_INVALID_STATE, "fh_color: no connectivity set");
    if (!c->has_host_conn) {  // mesh was given as device pointers: fetch the connectivity once
        std::vector<int> tmp(c->flat_len ? c->flat_len : 1);
        HIP_TRY(c, hipMemcpy(tmp.data(), c->conn.p, sizeof(int) * c->flat_len, hipMemcpyDeviceToHost));
        c->h_nodes.assign(tmp.begin(), tmp.begin() + c->flat_len);
        c->h_eoff.clear();
        c->has_host_conn = true;
    }
    std::vector<uint64_t> eoff_fixed;
    const uint64_t* eoff = c->h_eoff.data();
    if (!c->ragged) {
        eoff_fixed.resize(c->E + 1);
        for (uint64_t e = 0; e <= c->E; ++e) eoff_fixed[e] = e * (uint64_t)c->ei.n;
        eoff = eoff_fixed.data();
    }
    std::vector<uint64_t> offs, lab;
    static const uint64_t zero = 0;
    greedy_coloring(c->E, eoff, c->h_nodes.empty() ? &zero : c->h_nodes.data(), offs, lab);
    if (num_colors) *num_colors = offs.size() - 1;
    if (color_offsets) std::copy(offs.begin(), offs.end(), color_offsets);
    if (labels) std::copy(lab.begin(), lab.end(), labels);
    return upload_colors(c, offs, lab);
}

int fh_color_parallel(fh_ctx* c, uint64_t* num_colors, uint64_t* color_offsets, uint64_t* labels) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_mesh) return c->fail(FH_INVALID_STATE, "fh_color_parallel: no connectivity set");
    if (c->ragged) return c->fail(FH_UNSUPPORTED, "fh_color_parallel: ragged connectivity is coloured by fh_color (host)");
    if (c->E >= (1ull << 31)) return c->fail(FH_UNSUPPORTED, "fh_color_parallel: more than 2^31 elements");
    int rc = build_source_adjacency(c);
    if (rc) return rc;
    const int E = (int)c->E, n = c->ei.n;
    std::vector<uint64_t> offs(1, 0), lab((size_t)E);
    if (E > 0) {
        hipStream_t st = c->stream;
        DevBuf<int> color, tent, keys_s, flag;
        DevBuf<unsigned> ids, ids_s, remaining;
        HIP_TRY(c, color.alloc(E));
        HIP_TRY(c, tent.alloc(E));
        HIP_TRY(c, keys_s.alloc(E));
        HIP_TRY(c, ids.alloc(E));
        HIP_TRY(c, ids_s.alloc(E));
        HIP_TRY(c, flag.alloc(1));
        HIP_TRY(c, remaining.alloc(2));   // [0] elements still uncoloured after a round, [1] largest colour handed out
        HIP_TRY(c, hipMemsetAsync(flag.p, 0, sizeof(int), st));
        const int grid = (E + 255) / 256;
        hipLaunchKernelGGL(k_color_iota, dim3(grid), dim3(256), 0, st, E, ids.p, color.p, -1);
        HIP_TRY(c, hipMemsetAsync(remaining.p, 0, 2 * sizeof(unsigned), st));
        unsigned left = (unsigned)E, max_color = 0;
        int rounds = 0, over = 0;
        while (left > 0) {
            if (++rounds > 4096) return c->fail(FH_HIP_ERROR, "fh_color_parallel: no progress");
            HIP_TRY(c, hipMemsetAsync(remaining.p, 0, sizeof(unsigned), st));   // ([1] keeps its maximum over the rounds)
            hipLaunchKernelGGL(k_color_propose, dim3(grid), dim3(256), 0, st, E, n, c->conn.p, c->src_n2e_off.p, c->src_n2e.p, color.p, tent.p, flag.p);
            hipLaunchKernelGGL(k_color_resolve, dim3(grid), dim3(256), 0, st, E, n, c->conn.p, c->src_n2e_off.p, c->src_n2e.p, tent.p, color.p,
                               remaining.p);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(&left, remaining.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipMemcpyAsync(&over, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            if (over) return c->fail(FH_UNSUPPORTED, "fh_color_parallel: more than 32768 colours needed (use fh_color)");
        }
        HIP_TRY(c, hipMemcpyAsync(&max_color, remaining.p + 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        int key_bits = 1;
        while ((1u << key_bits) <= max_color) ++key_bits;
        // colours in order, the elements of a colour ascending: a stable sort of (colour, element)
        size_t tb = 0;
        HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, color.p, keys_s.p, ids.p, ids_s.p, E, 0, key_bits, st));
        DevBuf<char> tmp;
        HIP_TRY(c, tmp.alloc(tb + 16));
        HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, color.p, keys_s.p, ids.p, ids_s.p, E, 0, key_bits, st));
        std::vector<int> hk((size_t)E);
        std::vector<unsigned> hi((size_t)E);
        HIP_TRY(c, hipMemcpyAsync(hk.data(), keys_s.p, sizeof(int) * (size_t)E, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(hi.data(), ids_s.p, sizeof(unsigned) * (size_t)E, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        for (int i = 0; i < E; ++i) {
            lab[i] = hi[i];
            if (i > 0 && hk[i] != hk[i - 1]) offs.push_back((uint64_t)i);
        }
        offs.push_back((uint64_t)E);
        if (c->env("FENRIS_HIP_VERBOSE"))
            std::fprintf(stderr, "[fenris_hip] parallel colouring: %zu colours in %d rounds\n", offs.size() - 1, rounds);
    }
    if (num_colors) *num_colors = offs.size() - 1;
    if (color_offsets) std::copy(offs.begin(), offs.end(), color_offsets);
    if (labels) std::copy(lab.begin(), lab.end(), labels);
    return upload_colors(c, offs, lab);
}

int fh_set_colors(fh_ctx* c, uint64_t num_colors, const uint64_t* color_offsets, const uint64_t* labels) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_mesh) return c->fail(FH_INVALID_STATE, "fh_set_colors: no connectivity set");
    if (!color_offsets || (!labels && c->E)) return c->fail(FH_BAD_ARGUMENT, "fh_set_colors: null pointer");
    if (color_offsets[0] != 0 || color_offsets[num_colors] != c->E)
        return c->fail(FH_BAD_ARGUMENT, "fh_set_colors: offsets must cover all elements");
    std::vector<uint64_t> offs(color_offsets, color_offsets + num_colors + 1), lab(labels, labels + c->E);
    for (uint64_t e : lab)
        if (e >= c->E) return c->fail(FH_BAD_ARGUMENT, "fh_set_colors: label out of range");
    return upload_colors(c, offs, lab);
}

// ---- numeric assembly
}  // extern "C"
