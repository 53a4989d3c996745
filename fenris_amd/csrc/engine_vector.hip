// Residual, source vector, energy: launchers and C ABI
#include "engine_internal.hpp"

template <int EK, int OP>
static int launch_vector(fh_ctx* c, KArgs& a, size_t lds, int grid) {
    auto kern = k_assemble_vector<EK, OP>;
    if (lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}
template <int EK, int OP, int NT>
static int launch_vector_stream_nt(fh_ctx* c, KArgs& a) {
    constexpr int EPB = NT / ElemT<EK>::N;
    const size_t lds = make_layout<EK, OP, WHAT_VECTOR>(a.nq, EPB, 0, 0, false, 0, 1).bytes();
    if (lds > LDS_TARGET + 8 * 1024) return -1;
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
    const long long nbatch = (a.work_end - a.work_begin + EPB - 1) / EPB;
    const int per_cu = std::max(1, (int)std::min<size_t>(c->env_int("FENRIS_HIP_VEC_WGS_PER_CU", 3), (LDS_LIMIT - 512) / std::max<size_t>(lds, 1)));
    const int grid = std::max(1, (int)std::min<long long>(nbatch, (long long)c->env_int("FENRIS_HIP_PIPE_GRID", dev_cus * per_cu)));   // (tests force many batches per workgroup)
    auto kern = k_assemble_vector_stream<EK, OP, NT>;
    if (lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}
template <int EK, int OP>
static int launch_vector_stream(fh_ctx* c, KArgs& a) {
    if constexpr (ElemT<EK>::NG == ElemT<EK>::N && (ElemT<EK>::N == 4 || ElemT<EK>::N == 8)) {
        int rs = launch_vector_stream_nt<EK, OP, 256>(c, a);
        if (rs < 0) rs = launch_vector_stream_nt<EK, OP, 128>(c, a);
        return rs;
    } else {
        return -1;
    }
}
template <int EK, int OP>
static int launch_scalar(fh_ctx* c, KArgs& a, size_t lds, int grid) {
    auto kern = k_assemble_scalar<EK, OP>;
    if (lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

// register-resident element pass (element_pass.hpp): one thread per element of the small iso-parametric kinds, operators with a
// vector / scalar form.  Returns -1 when the combination is not covered (the callers keep the staged kernels).
template <int WHAT>
static int launch_element_pass(fh_ctx* c, KArgs& a) {
    const int grid = (int)((a.num_elements + 255) / 256);
    int rs = -1;
#define EP_OP(EKC)                                                                                                          \
    switch (c->op) {                                                                                                        \
        case FH_LAPLACE: hipLaunchKernelGGL((k_element_pass<EKC, FH_LAPLACE, WHAT>), dim3(grid), dim3(256), 0, c->stream, a); rs = FH_OK; break; \
        case FH_LINEAR_ELASTIC: hipLaunchKernelGGL((k_element_pass<EKC, FH_LINEAR_ELASTIC, WHAT>), dim3(grid), dim3(256), 0, c->stream, a); rs = FH_OK; break; \
        case FH_NEO_HOOKEAN: hipLaunchKernelGGL((k_element_pass<EKC, FH_NEO_HOOKEAN, WHAT>), dim3(grid), dim3(256), 0, c->stream, a); rs = FH_OK; break; \
        case FH_STVK: hipLaunchKernelGGL((k_element_pass<EKC, FH_STVK, WHAT>), dim3(grid), dim3(256), 0, c->stream, a); rs = FH_OK; break; \
        default: break;                                                                                                     \
    }
    switch (c->elem_kind) {
        case FH_QUAD4: EP_OP(FH_QUAD4) break;
        case FH_TRI3: EP_OP(FH_TRI3) break;
        case FH_TET4: EP_OP(FH_TET4) break;
        case FH_HEX8: EP_OP(FH_HEX8) break;
        default: break;
    }
#undef EP_OP
    if (rs == FH_OK) HIP_TRY(c, hipGetLastError());
    return rs;
}
static int launch_vector_from_elements_soa(fh_ctx* c, int sdim, const double* fe, double* out_dev, const unsigned* adj_off = nullptr,
                                           const unsigned* adj = nullptr, const SourceG* scaled = nullptr) {
    const int grid = (int)(((long long)c->N + 255) / 256);
    if (!adj_off) { adj_off = c->n2e_off.p; adj = c->n2e.p; }
    if (scaled) {   // scalar entries, sdim components g[c] sum
        if (sdim == 1) hipLaunchKernelGGL((k_vector_from_elements_soa<1, 1>), dim3(grid), dim3(256), 0, c->stream, (int)c->N, c->ei.n, (long long)c->E, adj_off, adj, fe, out_dev, *scaled);
        else if (sdim == 2) hipLaunchKernelGGL((k_vector_from_elements_soa<1, 2>), dim3(grid), dim3(256), 0, c->stream, (int)c->N, c->ei.n, (long long)c->E, adj_off, adj, fe, out_dev, *scaled);
        else hipLaunchKernelGGL((k_vector_from_elements_soa<1, 3>), dim3(grid), dim3(256), 0, c->stream, (int)c->N, c->ei.n, (long long)c->E, adj_off, adj, fe, out_dev, *scaled);
        HIP_TRY(c, hipGetLastError());
        return FH_OK;
    }
    if (sdim == 1) hipLaunchKernelGGL((k_vector_from_elements_soa<1, 0>), dim3(grid), dim3(256), 0, c->stream, (int)c->N, c->ei.n, (long long)c->E, adj_off, adj, fe, out_dev);
    else if (sdim == 2) hipLaunchKernelGGL((k_vector_from_elements_soa<2, 0>), dim3(grid), dim3(256), 0, c->stream, (int)c->N, c->ei.n, (long long)c->E, adj_off, adj, fe, out_dev);
    else hipLaunchKernelGGL((k_vector_from_elements_soa<3, 0>), dim3(grid), dim3(256), 0, c->stream, (int)c->N, c->ei.n, (long long)c->E, adj_off, adj, fe, out_dev);
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}
static bool element_pass_covers(const fh_ctx* c) {
    return !c->ragged && !c->env("FENRIS_HIP_NO_ELEMENT_PASS") &&
           (c->elem_kind == FH_HEX8 || c->elem_kind == FH_TET4 || c->elem_kind == FH_QUAD4 || c->elem_kind == FH_TRI3);
}

extern "C" {

static int assemble_vector_single(fh_ctx* c, double* out_dev, uint64_t* failed);
int fh_assemble_vector_dev(fh_ctx* c, double* out_dev, uint64_t* failed) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->rs.active) return assemble_vector_single(c, out_dev, failed);
    return rs_walk_accumulating(c, failed, [&](uint64_t* f) { return assemble_vector_single(c, out_dev, f); });
}
int fh_assemble_vector_async_dev(fh_ctx* c, double* out_dev) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    // nothing is read back between the groups of a rule-set table here, so a per-group reset would erase what an earlier group
    // reported: one reset in front of the walk (the device keeps the lowest failing element over all launches since the reset)
    if (c->rs.active) {
        const int r0 = reset_status(c);
        if (r0) return r0;
        c->keep_status = true;
    }
    c->defer_status = true;
    const int rc = fh_assemble_vector_dev(c, out_dev, nullptr);
    c->defer_status = false;
    c->keep_status = false;
    return rc;
}
// element tiles of the residual / source vector passes (vector_tiles.hip): once per mesh topology
static int ensure_vector_tiles(fh_ctx* c) {
    if (c->vt_gen == c->topo_gen) return FH_OK;
    int bad = 0;
    const hipError_t e = vector_tiles_build(c->stream, c->conn.p, c->ei.n, (long long)c->E, c->verts.p, c->ei.d, (int)c->N, &c->vt, &bad);
    if (e == hipErrorOutOfMemory) {   // no room for the tables: the callers keep the two-pass kernels
        (void)hipGetLastError();
        c->vt.release();
        bad = 1;
    } else {
        HIP_TRY(c, e);
    }
    c->vt_bad = bad != 0;
    c->vt_gen = c->topo_gen;
    return FH_OK;
}
static int assemble_vector_single(fh_ctx* c, double* out_dev, uint64_t* failed) {
    int rc = check_ready(c, "fh_assemble_vector", false);
    if (rc) return rc;
    if (c->op > FH_STVK) return c->fail(FH_UNSUPPORTED, "fh_assemble_vector: the mass assembler has no vector form");
    if (!out_dev) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_vector: out is null");
    rc = c->keep_status ? FH_OK : reset_status(c);
    if (rc) return rc;
    if (c->E == 0) return FH_OK;
    KArgs a;
    fill_common(c, a);
    a.vec_out = out_dev;
    a.work_begin = 0;
    a.work_end = (long long)(c->has_mask ? c->num_active : c->E);
    a.labels = c->has_mask ? c->active_list.p : nullptr;
    if (a.work_end == 0) return read_status(c, failed);
    // small iso-parametric elements: tiles of 256 elements, one thread per element, the tile's distinct nodes summed in LDS, only
    // those partial sums through HBM, then one thread per node (vector_tiles.hip); no atomics, bitwise reproducible; an element mask
    // zeroes the contributions of the inactive elements
    if (element_pass_covers(c) && !c->env("FENRIS_HIP_VECTOR_ATOMICS") && !c->env("FENRIS_HIP_NO_VECTOR_TILES") && c->op <= FH_STVK) {
        rc = ensure_vector_tiles(c);
        if (rc) return rc;
        if (!c->vt_bad) {
            const size_t need = (size_t)c->vt.v.npartials * c->S();
            if (c->fe_scratch.n < need) HIP_TRY(c, c->fe_scratch.alloc(need));
            KArgs at = a;
            at.labels = nullptr;
            const int rs = vector_tiles_element_pass(c->elem_kind, c->op, c->stream, at, c->vt.v, c->has_mask ? c->active.p : nullptr, c->fe_scratch.p);
            if (rs == FH_OK) {
                HIP_TRY(c, hipGetLastError());
                c->last_kernel = "k_element_pass_tiled + k_vector_from_partials";
                HIP_TRY(c, vector_tiles_node_pass(c->stream, c->S(), (int)c->N, c->vt.v, c->fe_scratch.p, out_dev));
                return read_status(c, failed);
            }
        }
    }
    // small iso-parametric elements without an element list: one thread per element, element vectors laid out by local node, then
    // one thread per node (element_pass.hpp); no atomics, bitwise reproducible
    if (!a.labels && element_pass_covers(c) && !c->env("FENRIS_HIP_VECTOR_ATOMICS")) {
        rc = build_pattern(c);  // the node -> (element, local node) adjacency comes with the pattern
        if (rc) return rc;
        const size_t need = (size_t)c->E * c->ei.n * c->S();
        if (c->fe_scratch.n < need) HIP_TRY(c, c->fe_scratch.alloc(need));
        a.ke_out = c->fe_scratch.p;
        const int rs = launch_element_pass<EP_VECTOR>(c, a);
        if (rs == FH_OK) {
            c->last_kernel = "k_element_pass + k_vector_from_elements_soa";
            rc = launch_vector_from_elements_soa(c, c->S(), c->fe_scratch.p, out_dev);
            if (rc) return rc;
            return read_status(c, failed);
        }
        if (rs > 0) return rs;
        a.ke_out = nullptr;
    }
    // persistent, prefetching form for the small iso-parametric elements (no element list: a mask keeps the generic kernel).
    // Two passes by default: element vectors to a scratch buffer, then one thread per row sums its node's entries in
    // ascending element order -- no atomics, bitwise reproducible (FENRIS_HIP_VECTOR_ATOMICS keeps the one-pass scatter)
    if (!a.labels) {
        const bool two_pass = !c->env("FENRIS_HIP_VECTOR_ATOMICS") && !c->ragged &&
                              (c->elem_kind == FH_HEX8 || c->elem_kind == FH_TET4 || c->elem_kind == FH_QUAD4);
        if (two_pass) {
            rc = build_pattern(c);  // the node -> (element, local node) adjacency comes with the pattern
            if (rc) return rc;
            const size_t need = (size_t)c->E * c->ei.n * c->S();
            if (c->fe_scratch.n < need) HIP_TRY(c, c->fe_scratch.alloc(need));
            a.ke_out = c->fe_scratch.p;
        }
        int rs = -1;
#define CALL(EKC, OPC) rs = launch_vector_stream<EKC, OPC>(c, a)
        FH_FOR_ELEM_OP(c->elem_kind, c->op, CALL)
#undef CALL
        if (rs == FH_OK && two_pass) {
            const long long rows = (long long)c->N * c->S();
            const int grid = (int)((rows + 255) / 256);
            if (c->S() == 1) hipLaunchKernelGGL(k_vector_from_elements<1>, dim3(grid), dim3(256), 0, c->stream, (int)c->N, c->n2e_off.p, c->n2e.p, c->fe_scratch.p, out_dev);
            else if (c->S() == 2) hipLaunchKernelGGL(k_vector_from_elements<2>, dim3(grid), dim3(256), 0, c->stream, (int)c->N, c->n2e_off.p, c->n2e.p, c->fe_scratch.p, out_dev);
            else hipLaunchKernelGGL(k_vector_from_elements<3>, dim3(grid), dim3(256), 0, c->stream, (int)c->N, c->n2e_off.p, c->n2e.p, c->fe_scratch.p, out_dev);
            HIP_TRY(c, hipGetLastError());
        }
        if (rs == FH_OK) return read_status(c, failed);
        if (rs > 0) return rs;
        a.ke_out = nullptr;
    }
    a.epb = choose_epb(c, WHAT_VECTOR);
    a.ub = a.epb;
    const size_t lds = layout_bytes_dyn(c->elem_kind, c->op, WHAT_VECTOR, c->nq, a.ub, 0, 0, false);
    if (lds > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "quadrature rule too large for LDS staging");
    const int grid = (int)((a.work_end + a.epb - 1) / a.epb);
#define CALL(EKC, OPC) rc = launch_vector<EKC, OPC>(c, a, lds, grid)
    FH_FOR_ELEM_OP(c->elem_kind, c->op, CALL)
#undef CALL
    if (rc) return rc;
    return read_status(c, failed);
}

int fh_assemble_vector(fh_ctx* c, double* out, uint64_t* failed) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = check_ready(c, "fh_assemble_vector", false);
    if (rc) return rc;
    if (!out) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_vector: out is null");
    const size_t len = (size_t)c->S() * c->N;
    DevBuf<double> d;
    HIP_TRY(c, d.alloc(len));
    HIP_TRY(c, hipMemcpyAsync(d.p, out, sizeof(double) * len, hipMemcpyHostToDevice, c->stream));
    rc = fh_assemble_vector_dev(c, d.p, failed);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, d.p, sizeof(double) * len, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FH_OK;
}

// ---- ElementSourceAssembler (src/assembly/local/source.rs) ------------------------------------------------------
extern "C++" int source_ready(fh_ctx* c, const char* who) {
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, std::string(who) + ": no finite element mesh set");
    if (c->nq <= 0) return c->fail(FH_INVALID_STATE, std::string(who) + ": no quadrature table set");
    return FH_OK;
}

int fh_assemble_source_vector_dev(fh_ctx* c, uint32_t sdim, const double* g, const double* values_dev, double* out_dev) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (c->rs.active) return c->fail(FH_UNSUPPORTED, "fh_assemble_source_vector: rule-set quadrature tables (fh_set_quadrature_rules) are not walked here");
    int rc = source_ready(c, "fh_assemble_source_vector");
    if (rc) return rc;
    const int D = c->ei.d;
    if (!out_dev) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_source_vector: out is null");
    if (sdim != 1 && (int)sdim != D) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_source_vector: solution dim must be 1 or the geometry dim");
    if (!values_dev && !g) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_source_vector: neither g nor values given");
    if (!values_dev && !c->has_params)
        return c->fail(FH_INVALID_STATE, "fh_assemble_source_vector: the uniform source needs the density in the quadrature table");
    if (c->E == 0) return FH_OK;
    KArgs a;
    fill_common(c, a);
    SourceArgs sa{};
    sa.N = c->ei.n;
    sa.NG = c->ei.ng;
    sa.phigeom = c->phigeom.p;
    sa.values = values_dev;
    DevBuf<double> gd;      // device copy of g: only the one-pass scatter below reads it through a pointer
    SourceG gval{{0.0, 0.0, 0.0}};
    if (!values_dev)
        for (uint32_t k = 0; k < sdim; ++k) gval.v[k] = g[k];
    a.vec_out = out_dev;
    a.work_begin = 0;
    a.work_end = (long long)(c->has_mask ? c->num_active : c->E);
    a.labels = c->has_mask ? c->active_list.p : nullptr;
    if (a.work_end == 0) return FH_OK;
    // small iso-parametric elements: the tiles of the residual (vector_tiles.hip) -- element vectors summed per distinct node of a tile
    // in LDS, partial sums through HBM, one thread per node; an element mask zeroes the inactive elements
    if (element_pass_covers(c) && !c->env("FENRIS_HIP_VECTOR_ATOMICS") && !c->env("FENRIS_HIP_NO_VECTOR_TILES")) {
        rc = ensure_vector_tiles(c);
        if (rc) return rc;
        if (!c->vt_bad) {
            const bool fact = !values_dev;   // GravitySource: scalar partials, the node sum multiplies by g
            const size_t need = (size_t)c->vt.v.npartials * (fact ? 1 : sdim);
            if (c->fe_scratch.n < need) HIP_TRY(c, c->fe_scratch.alloc(need));
            KArgs at = a;
            at.labels = nullptr;
            if (vector_tiles_source_pass(D, (int)sdim, c->ei.n, fact, c->stream, at, gval.v, sa.values, c->vt.v, c->has_mask ? c->active.p : nullptr,
                                         c->fe_scratch.p) == 0) {
                HIP_TRY(c, hipGetLastError());
                c->last_kernel = "k_source_elements_tiled + k_vector_from_partials";
                HIP_TRY(c, vector_tiles_node_pass(c->stream, (int)sdim, (int)c->N, c->vt.v, c->fe_scratch.p, out_dev, fact ? gval.v : nullptr));
                return FH_OK;
            }
        }
    }
    // two passes without atomics where the node adjacency is available (it comes with the pattern, which needs an operator
    // for the solution dimension): element vectors to scratch, then a per-row sum in element order
    bool two_pass = !a.labels && !c->ragged && c->op >= 0 && !c->env("FENRIS_HIP_VECTOR_ATOMICS");
    if (two_pass && build_pattern(c) != FH_OK) two_pass = false;
    // a context without an operator (the usual case of a source assembler): the adjacency alone, for the element pass
    const unsigned *adj_off = nullptr, *adj = nullptr;
    if (!two_pass && !a.labels && c->op < 0 && element_pass_covers(c) && !c->env("FENRIS_HIP_VECTOR_ATOMICS") && build_source_adjacency(c) == FH_OK) {
        two_pass = true;
        adj_off = c->src_n2e_off.p;
        adj = c->src_n2e.p;
    }
    if (two_pass) {
        const size_t need = (size_t)c->E * c->ei.n * sdim;
        if (c->fe_scratch.n < need) HIP_TRY(c, c->fe_scratch.alloc(need));
        a.ke_out = c->fe_scratch.p;
    }
    if (two_pass && element_pass_covers(c)) {   // one thread per element, element vectors by local node, one thread per node (element_pass.hpp)
        const int ge = (int)((c->E + 255) / 256);
        double* fe = c->fe_scratch.p;
        const bool fact = !values_dev;   // GravitySource: scalar element entries, the node sum multiplies by g (element_pass.hpp)
#define SRC(DV, SV, NV)                                                                                                                     \
        do {                                                                                                                                \
            if (fact) hipLaunchKernelGGL((k_source_elements<DV, SV, NV, true>), dim3(ge), dim3(256), 0, c->stream, a, gval, sa.values, fe); \
            else hipLaunchKernelGGL((k_source_elements<DV, SV, NV, false>), dim3(ge), dim3(256), 0, c->stream, a, gval, sa.values, fe);     \
        } while (0)
        const int n = c->ei.n;
        if (D == 2 && n == 4) { if (sdim == 1) SRC(2, 1, 4); else SRC(2, 2, 4); }
        else if (D == 2) { if (sdim == 1) SRC(2, 1, 3); else SRC(2, 2, 3); }
        else if (n == 8) { if (sdim == 1) SRC(3, 1, 8); else SRC(3, 3, 8); }
        else { if (sdim == 1) SRC(3, 1, 4); else SRC(3, 3, 4); }
#undef SRC
        HIP_TRY(c, hipGetLastError());
        c->last_kernel = "k_source_elements + k_vector_from_elements_soa";
        return launch_vector_from_elements_soa(c, (int)sdim, fe, out_dev, adj_off, adj, fact ? &gval : nullptr);
    }
    if (adj_off) { two_pass = false; a.ke_out = nullptr; }   // (not covered after all: the one-pass scatter)
    if (!values_dev) {
        HIP_TRY(c, gd.alloc(sdim));
        HIP_TRY(c, hipMemcpyAsync(gd.p, g, sizeof(double) * sdim, hipMemcpyHostToDevice, c->stream));
        sa.g = gd.p;
    }
    a.epb = std::max(1, 256 / std::max(c->nq, c->ei.n));
    const size_t lds = sizeof(double) * (size_t)a.epb * c->nq;
    const int grid = (int)((a.work_end + a.epb - 1) / a.epb);
    c->last_kernel = "k_assemble_source";
    if (D == 2 && sdim == 1) hipLaunchKernelGGL((k_assemble_source<2, 1>), dim3(grid), dim3(256), lds, c->stream, a, sa);
    else if (D == 2) hipLaunchKernelGGL((k_assemble_source<2, 2>), dim3(grid), dim3(256), lds, c->stream, a, sa);
    else if (sdim == 1) hipLaunchKernelGGL((k_assemble_source<3, 1>), dim3(grid), dim3(256), lds, c->stream, a, sa);
    else hipLaunchKernelGGL((k_assemble_source<3, 3>), dim3(grid), dim3(256), lds, c->stream, a, sa);
    HIP_TRY(c, hipGetLastError());
    if (two_pass) {
        const long long rows = (long long)c->N * sdim;
        const int g2 = (int)((rows + 255) / 256);
        if (sdim == 1) hipLaunchKernelGGL(k_vector_from_elements<1>, dim3(g2), dim3(256), 0, c->stream, (int)c->N, c->n2e_off.p, c->n2e.p, c->fe_scratch.p, out_dev);
        else if (sdim == 2) hipLaunchKernelGGL(k_vector_from_elements<2>, dim3(g2), dim3(256), 0, c->stream, (int)c->N, c->n2e_off.p, c->n2e.p, c->fe_scratch.p, out_dev);
        else hipLaunchKernelGGL(k_vector_from_elements<3>, dim3(g2), dim3(256), 0, c->stream, (int)c->N, c->n2e_off.p, c->n2e.p, c->fe_scratch.p, out_dev);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // gd is released on return
    return FH_OK;
}

int fh_assemble_source_vector(fh_ctx* c, uint32_t sdim, const double* g, const double* values, double* out) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = source_ready(c, "fh_assemble_source_vector");
    if (rc) return rc;
    if (!out) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_source_vector: out is null");
    const size_t len = (size_t)sdim * c->N, nv = (size_t)c->E * c->nq * sdim;
    DevBuf<double> d, v;
    HIP_TRY(c, d.alloc(len + 1));
    HIP_TRY(c, hipMemcpyAsync(d.p, out, sizeof(double) * len, hipMemcpyHostToDevice, c->stream));
    if (values) {
        HIP_TRY(c, v.alloc(nv + 1));
        HIP_TRY(c, hipMemcpyAsync(v.p, values, sizeof(double) * nv, hipMemcpyHostToDevice, c->stream));
    }
    rc = fh_assemble_source_vector_dev(c, sdim, g, values ? v.p : nullptr, d.p);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, d.p, sizeof(double) * len, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FH_OK;
}

int fh_physical_quadrature_points_dev(fh_ctx* c, double* x_dev) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (c->rs.active) return c->fail(FH_UNSUPPORTED, "fh_physical_quadrature_points: rule-set quadrature tables (fh_set_quadrature_rules) are not walked here");
    int rc = source_ready(c, "fh_physical_quadrature_points");
    if (rc) return rc;
    if (!x_dev) return c->fail(FH_BAD_ARGUMENT, "fh_physical_quadrature_points: output is null");
    if (c->E == 0) return FH_OK;
    KArgs a;
    fill_common(c, a);
    SourceArgs sa{};
    sa.N = c->ei.n;
    sa.NG = c->ei.ng;
    sa.phigeom = c->phigeom.p;
    sa.xq = x_dev;
    const long long total = (long long)c->E * c->nq;
    const int grid = (int)((total + 255) / 256);
    if (c->ei.d == 2) hipLaunchKernelGGL((k_physical_points<2>), dim3(grid), dim3(256), 0, c->stream, a, sa);
    else hipLaunchKernelGGL((k_physical_points<3>), dim3(grid), dim3(256), 0, c->stream, a, sa);
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

int fh_physical_quadrature_points(fh_ctx* c, double* x) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = source_ready(c, "fh_physical_quadrature_points");
    if (rc) return rc;
    if (!x) return c->fail(FH_BAD_ARGUMENT, "fh_physical_quadrature_points: output is null");
    const size_t n = (size_t)c->E * c->nq * c->ei.d;
    DevBuf<double> d;
    HIP_TRY(c, d.alloc(n + 1));
    rc = fh_physical_quadrature_points_dev(c, d.p);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(x, d.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FH_OK;
}

static int assemble_scalar_single(fh_ctx* c, double* out, uint64_t* failed);
int fh_assemble_scalar(fh_ctx* c, double* out, uint64_t* failed) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->rs.active) return assemble_scalar_single(c, out, failed);
    if (!out) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_scalar: out is null");
    double tot = 0.0;
    const int rc = rs_walk_accumulating(c, failed, [&](uint64_t* f) {
        double part = 0.0;
        const int r = assemble_scalar_single(c, &part, f);
        tot += part;
        return r;
    });
    *out = tot;
    return rc;
}
static int assemble_scalar_single(fh_ctx* c, double* out, uint64_t* failed) {
    int rc = check_ready(c, "fh_assemble_scalar", false);
    if (rc) return rc;
    if (c->op > FH_STVK) return c->fail(FH_UNSUPPORTED, "fh_assemble_scalar: the mass assembler has no scalar form");
    if (!out) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_scalar: out is null");
    rc = reset_status(c);
    if (rc) return rc;
    *out = 0.0;
    if (c->E == 0) return FH_OK;
    KArgs a;
    fill_common(c, a);
    a.work_begin = 0;
    a.work_end = (long long)(c->has_mask ? c->num_active : c->E);
    a.labels = c->has_mask ? c->active_list.p : nullptr;
    if (a.work_end == 0) return FH_OK;
    // element tiles (vector_tiles.hip): the elements in the tiles' (space-compact) order -- what makes the gathers local on a numbering
    // without locality (C3's permuted tetrahedra: 0.76 -> 0.20 ms per call); an element mask zeroes the inactive elements' energies
    if (element_pass_covers(c) && !c->env("FENRIS_HIP_NO_VECTOR_TILES")) {
        rc = ensure_vector_tiles(c);
        if (rc) return rc;
        if (!c->vt_bad) {
            const int grid = vector_tiles_energy_partials(c->vt.v);
            if (c->scalar_partial.n < (size_t)grid + 1) HIP_TRY(c, c->scalar_partial.alloc((size_t)grid + 1));
            KArgs at = a;
            at.labels = nullptr;
            if (vector_tiles_energy_pass(c->elem_kind, c->op, c->stream, at, c->vt.v, c->has_mask ? c->active.p : nullptr, c->scalar_partial.p) == grid) {
                HIP_TRY(c, hipGetLastError());
                c->last_kernel = "k_element_energy_tiled";
                hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->scalar_partial.p, grid, c->scalar_partial.p + grid);
                HIP_TRY(c, hipGetLastError());
                HIP_TRY(c, hipMemcpyAsync(out, c->scalar_partial.p + grid, sizeof(double), hipMemcpyDeviceToHost, c->stream));
                return read_status(c, failed);
            }
        }
    }
    if (!a.labels && element_pass_covers(c)) {
        // one thread per element (element_pass.hpp), workgroup partials in a fixed tree, the partials summed in index order by one
        // workgroup: one double comes back (global.rs:703-709 sums element by element; same terms, fixed association)
        const int grid = (int)((c->E + 255) / 256);
        DevBuf<double> partial;
        HIP_TRY(c, partial.alloc((size_t)grid + 1));
        a.scalar_out = partial.p;
        const int rs = launch_element_pass<EP_SCALAR>(c, a);
        if (rs == FH_OK) {
            c->last_kernel = "k_element_pass<scalar>";
            hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, partial.p, grid, partial.p + grid);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(out, partial.p + grid, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            return read_status(c, failed);
        }
        if (rs > 0) return rs;
    }
    // a batch of elements per workgroup: element energies summed in element order inside the batch, the batch partials in
    // order on the host (global.rs:703-709 sums element by element; same terms, fixed association)
    a.epb = std::max(1, std::min(choose_epb(c, WHAT_SCALAR), std::max(1, 256 / std::max(c->nq, 1))));
    a.ub = a.epb;
    const size_t lds = layout_bytes_dyn(c->elem_kind, c->op, WHAT_SCALAR, c->nq, a.ub, 0, 0, false);
    const int grid = (int)((a.work_end + a.epb - 1) / a.epb);
    DevBuf<double> partial;
    HIP_TRY(c, partial.alloc((size_t)grid));
    a.scalar_out = partial.p;
#define CALL(EKC, OPC) rc = launch_scalar<EKC, OPC>(c, a, lds, grid)
    FH_FOR_ELEM_OP(c->elem_kind, c->op, CALL)
#undef CALL
    if (rc) return rc;
    std::vector<double> h((size_t)grid);
    HIP_TRY(c, hipMemcpyAsync(h.data(), partial.p, sizeof(double) * grid, hipMemcpyDeviceToHost, c->stream));
    rc = read_status(c, failed);
    if (rc) return rc;
    double tot = 0.0;
    for (double v : h) tot += v;
    *out = tot;
    return FH_OK;
}

// ---- Dirichlet helpers
}  // extern "C"
