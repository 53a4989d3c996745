// Dirichlet rows, SpMV, Jacobi-PCG, error integrals: C ABI
#include "engine_internal.hpp"

extern "C" {
int fh_apply_dirichlet_csr_dev(fh_ctx* c, double* values_dev, const uint64_t* nodes, uint64_t n) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_pattern) return c->fail(FH_INVALID_STATE, "fh_apply_dirichlet_csr_dev: call fh_pattern first");
    if (!values_dev || (n && !nodes)) return c->fail(FH_BAD_ARGUMENT, "fh_apply_dirichlet_csr_dev: null pointer");
    const int S = c->S(), N = (int)c->N;
    for (uint64_t i = 0; i < n; ++i)
        if (nodes[i] >= c->N) return c->fail(FH_BAD_ARGUMENT, "Dirichlet node out of range");
    // membership flags on the device from the node list (round 4: a host array of N bytes filled and uploaded per call, an entry-wise
    // kernel that searched each entry's row by bisection and a host round trip for the scale made this step 11 ms on the 216^3 mesh)
    DevBuf<unsigned char> dm;
    DevBuf<unsigned long long> first, dn;
    DevBuf<double> scale;
    HIP_TRY(c, dm.alloc((size_t)N + 1));
    HIP_TRY(c, first.alloc(1));
    HIP_TRY(c, scale.alloc(1));
    HIP_TRY(c, dn.alloc((size_t)n + 1));
    HIP_TRY(c, hipMemsetAsync(dm.p, 0, (size_t)N + 1, c->stream));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(dn.p, nodes, sizeof(uint64_t) * n, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_mark_nodes, dim3(grid_for((long long)n, 256, 1 << 30)), dim3(256), 0, c->stream, dn.p, (long long)n, dm.p);
    }
    HIP_TRY(c, hipMemsetAsync(first.p, 0xff, sizeof(unsigned long long), c->stream));
    const long long R = (long long)N * S;
    hipLaunchKernelGGL(k_first_nonzero_diag, dim3(grid_for(R, 256, 1 << 30)), dim3(256), 0, c->stream, c->noff.p, c->ncols.p, N, S,
                       values_dev, first.p, (double*)nullptr);
    hipLaunchKernelGGL(k_first_nonzero_diag, dim3(1), dim3(64), 0, c->stream, c->noff.p, c->ncols.p, N, S, values_dev, first.p,
                       scale.p);
    if (c->nnz_nodes)
        hipLaunchKernelGGL(k_dirichlet_rows, dim3(grid_for(((long long)N + 7) / 8, 1, 1 << 20)), dim3(256), 0, c->stream, c->noff.p,
                           c->ncols.p, N, S, dm.p, values_dev, scale.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // (the temporaries are released on return)
    return FH_OK;
}

int fh_apply_dirichlet_rhs_dev(fh_ctx* c, double* rhs_dev, const uint64_t* nodes, uint64_t n) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!rhs_dev || (n && !nodes)) return c->fail(FH_BAD_ARGUMENT, "fh_apply_dirichlet_rhs_dev: null pointer");
    if (n == 0) return FH_OK;
    for (uint64_t i = 0; i < n; ++i)
        if (nodes[i] >= c->N) return c->fail(FH_BAD_ARGUMENT, "Dirichlet node out of range");
    const int S = c->S();
    DevBuf<unsigned long long> dn;
    HIP_TRY(c, dn.alloc((size_t)n));
    HIP_TRY(c, hipMemcpyAsync(dn.p, nodes, sizeof(uint64_t) * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_dirichlet_rhs, dim3(grid_for((long long)n * S, 256, 1 << 30)), dim3(256), 0, c->stream, rhs_dev, dn.p,
                       (long long)n, S);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

// ---- callers that keep K on the device: CG and the error integrals ---------------------------------
// sum of per-workgroup partials (stride K) in workgroup order: deterministic
static int sum_partials(fh_ctx* c, const double* dev, int blocks, int K, double* out) {
    std::vector<double> h((size_t)blocks * K);
    HIP_TRY(c, hipMemcpyAsync(h.data(), dev, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int b = 0; b < blocks; ++b) s += h[(size_t)b * K + k];
        out[k] = s;
    }
    return FH_OK;
}

// y = A x (+ the partial sums of x . y, `partials` of them, when `partial` is given).  ONE trip per wavefront: as many short-lived workgroups as the rows need,
// dispatched in order, instead of a resident grid striding over the rows -- the rows in flight stay one compact window of the value array (Hex8
// elasticity 216^3, 19.7 GB of values: 4 096 striding workgroups 5.34 ms, 32 768: 4.99, one trip each (636 k): 4.65; 1 280 = exactly the resident
// ones: 6.7).  The per-workgroup partials of x . y go to `scratch` and are summed over `partials` contiguous ranges in a fixed order.
static int spmv_launch(fh_ctx* c, const double* vals, const double* x, double* y, double* partial, int partials, DevBuf<double>* scratch) {
    const int N = (int)c->N;
    const bool half = c->max_row <= 32 && !c->env("FENRIS_HIP_SPMV_WAVE_PER_NODE");   // half a wavefront per node, one lane per column block
    const int grid = std::max(1, half ? (N + 15) / 16 : (N + 3) / 4);
    double* wg_partial = nullptr;
    if (partial) {
        if (scratch->n < (size_t)grid) HIP_TRY(c, scratch->alloc((size_t)grid));
        wg_partial = scratch->p;
    }
    if (half) {
        switch (c->S()) {
            case 1: hipLaunchKernelGGL((k_spmv_blocked_half<1>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, wg_partial); break;
            case 2: hipLaunchKernelGGL((k_spmv_blocked_half<2>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, wg_partial); break;
            default:
                hipLaunchKernelGGL((k_spmv_blocked_half<3>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, wg_partial);
                break;
        }
    } else {
        switch (c->S()) {
            case 1: hipLaunchKernelGGL((k_spmv_blocked<1>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, wg_partial); break;
            case 2: hipLaunchKernelGGL((k_spmv_blocked<2>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, wg_partial); break;
            default: hipLaunchKernelGGL((k_spmv_blocked<3>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, wg_partial); break;
        }
    }
    HIP_TRY(c, hipGetLastError());
    if (partial) {
        hipLaunchKernelGGL(k_sum_partial_ranges<1>, dim3(partials), dim3(256), 0, c->stream, wg_partial, (long long)grid, partial);
        HIP_TRY(c, hipGetLastError());
    }
    return FH_OK;
}
static int matrix_ready(fh_ctx* c, const char* who) {
    if (!c->has_pattern) return c->fail(FH_INVALID_STATE, std::string(who) + ": call fh_pattern first");
    if (c->S() < 1 || c->S() > 3) return c->fail(FH_UNSUPPORTED, std::string(who) + ": solution dim must be 1..3");
    return FH_OK;
}

int fh_spmv_dev(fh_ctx* c, const double* values_dev, const double* x_dev, double* y_dev) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = matrix_ready(c, "fh_spmv");
    if (rc) return rc;
    if (!values_dev || !x_dev || !y_dev) return c->fail(FH_BAD_ARGUMENT, "fh_spmv: null argument");
    if (c->N == 0) return FH_OK;
    c->last_kernel = (c->max_row <= 32 && !c->env("FENRIS_HIP_SPMV_WAVE_PER_NODE")) ? "k_spmv_blocked_half" : "k_spmv_blocked";
    return spmv_launch(c, values_dev, x_dev, y_dev, nullptr, 0, nullptr);
}

int fh_cg_solve_dev(fh_ctx* c, const double* values_dev, const double* b_dev, double* x_dev, int preconditioner, double rel_tol,
                    uint64_t max_iter, uint64_t* num_iterations) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (num_iterations) *num_iterations = 0;
    int rc = matrix_ready(c, "fh_cg_solve");
    if (rc) return rc;
    if (!values_dev || !b_dev || !x_dev) return c->fail(FH_BAD_ARGUMENT, "fh_cg_solve: null argument");
    if (preconditioner != FH_PRECOND_IDENTITY && preconditioner != FH_PRECOND_JACOBI)
        return c->fail(FH_BAD_ARGUMENT, "fh_cg_solve: unknown preconditioner");
    const int S = c->S();
    const int n = S * (int)c->N;
    if (n == 0) return FH_OK;
    const int gv = std::min(1024, (n + 255) / 256);                               // vector kernels: ranges of their per-workgroup partials
    const int gvb = std::max(1, (n + 255) / 256);                                 // ... and their workgroups: one entry per thread, short-lived (see spmv_launch)
    const int gs = (int)std::min<uint64_t>(2048, (c->N + 3) / 4);                  // SpMV: ranges of its per-workgroup partials
    DevBuf<double> r, z, p, Ap, dinv, partial, wg_partial;
    HIP_TRY(c, r.alloc(n));
    HIP_TRY(c, z.alloc(n));
    HIP_TRY(c, p.alloc(n));
    HIP_TRY(c, Ap.alloc(n));
    HIP_TRY(c, partial.alloc((size_t)3 * std::max(gv, gs)));
    HIP_TRY(c, wg_partial.alloc((size_t)3 * gvb));
    if (preconditioner == FH_PRECOND_JACOBI) {
        HIP_TRY(c, dinv.alloc(n));
        const int g = (n + 255) / 256;
        switch (S) {
            case 1: hipLaunchKernelGGL((k_inverse_diagonal<1>), dim3(g), dim3(256), 0, c->stream, (int)c->N, c->noff.p, c->ncols.p, values_dev, dinv.p); break;
            case 2: hipLaunchKernelGGL((k_inverse_diagonal<2>), dim3(g), dim3(256), 0, c->stream, (int)c->N, c->noff.p, c->ncols.p, values_dev, dinv.p); break;
            default: hipLaunchKernelGGL((k_inverse_diagonal<3>), dim3(g), dim3(256), 0, c->stream, (int)c->N, c->noff.p, c->ncols.p, values_dev, dinv.p); break;
        }
        HIP_TRY(c, hipGetLastError());
    }
    c->last_kernel = (c->max_row <= 32 && !c->env("FENRIS_HIP_SPMV_WAVE_PER_NODE")) ? "k_spmv_blocked_half" : "k_spmv_blocked";
    // r = b - A x;  z = P r;  p = z   (cg.rs:388-404)
    rc = spmv_launch(c, values_dev, x_dev, r.p, nullptr, 0, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cg_init, dim3(gvb), dim3(256), 0, c->stream, n, b_dev, dinv.p, r.p, z.p, p.p, wg_partial.p);
    hipLaunchKernelGGL(k_sum_partial_ranges<3>, dim3(gv), dim3(256), 0, c->stream, wg_partial.p, (long long)gvb, partial.p);
    HIP_TRY(c, hipGetLastError());
    double s3[3];
    rc = sum_partials(c, partial.p, gv, 3, s3);
    if (rc) return rc;
    double zTr = s3[0];
    const double b_norm = std::sqrt(s3[1]);
    double r_norm = std::sqrt(s3[2]);
    if (b_norm == 0.0) {  // cg.rs:409-412
        HIP_TRY(c, hipMemsetAsync(x_dev, 0, sizeof(double) * (size_t)n, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return FH_OK;
    }
    uint64_t it = 0;
    int status = FH_OK;
    for (;;) {
        if (r_norm <= rel_tol * b_norm) break;  // RelativeResidualCriterion, cg.rs:108-124
        if (max_iter && it >= max_iter) { status = FH_CG_MAX_ITERATIONS; break; }
        double pAp;
        rc = spmv_launch(c, values_dev, p.p, Ap.p, partial.p, gs, &wg_partial);
        if (rc) return rc;
        rc = sum_partials(c, partial.p, gs, 1, &pAp);
        if (rc) return rc;
        if (pAp <= 0.0) { status = FH_CG_INDEFINITE_OPERATOR; break; }
        if (zTr <= 0.0) { status = FH_CG_INDEFINITE_PRECONDITIONER; break; }
        const double alpha = zTr / pAp;
        hipLaunchKernelGGL(k_cg_update, dim3(gvb), dim3(256), 0, c->stream, n, alpha, p.p, Ap.p, dinv.p, x_dev, r.p, z.p, wg_partial.p);
        hipLaunchKernelGGL(k_sum_partial_ranges<2>, dim3(gv), dim3(256), 0, c->stream, wg_partial.p, (long long)gvb, partial.p);
        HIP_TRY(c, hipGetLastError());
        ++it;
        double s2[2];
        rc = sum_partials(c, partial.p, gv, 2, s2);
        if (rc) return rc;
        const double beta = s2[0] / zTr;
        r_norm = std::sqrt(s2[1]);
        hipLaunchKernelGGL(k_cg_direction, dim3(gvb), dim3(256), 0, c->stream, n, beta, z.p, p.p);
        HIP_TRY(c, hipGetLastError());
        zTr = s2[0];
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (num_iterations) *num_iterations = it;
    if (status == FH_CG_MAX_ITERATIONS) return c->fail(status, "CG: max iterations reached");
    if (status == FH_CG_INDEFINITE_OPERATOR) return c->fail(status, "CG: operator appears to be indefinite");
    if (status == FH_CG_INDEFINITE_PRECONDITIONER) return c->fail(status, "CG: indefinite preconditioner");
    return FH_OK;
}

int fh_cg_solve(fh_ctx* c, const double* values, const double* b, double* x, int preconditioner, double rel_tol, uint64_t max_iter,
                uint64_t* num_iterations) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = matrix_ready(c, "fh_cg_solve");
    if (rc) return rc;
    if (!values || !b || !x) return c->fail(FH_BAD_ARGUMENT, "fh_cg_solve: null argument");
    const size_t n = (size_t)c->S() * c->N, nnz = (size_t)c->S() * c->S() * c->nnz_nodes;
    DevBuf<double> dv, db, dx;
    HIP_TRY(c, dv.alloc(nnz + 1));
    HIP_TRY(c, db.alloc(n + 1));
    HIP_TRY(c, dx.alloc(n + 1));
    HIP_TRY(c, hipMemcpyAsync(dv.p, values, sizeof(double) * nnz, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(db.p, b, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    rc = fh_cg_solve_dev(c, dv.p, db.p, dx.p, preconditioner, rel_tol, max_iter, num_iterations);
    // like the reference's SolveError, the iterate reached so far is handed back on failure
    HIP_TRY(c, hipMemcpyAsync(x, dx.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return rc;
}

static int error_squared(fh_ctx* c, int which, uint32_t sdim, const double* uh_dev, const double* exact_dev, double* out) {
    DevGuard dev_guard_(c->device);
    if (c->rs.active) return c->fail(FH_UNSUPPORTED, "fh_estimate_*_error_squared: rule-set quadrature tables (fh_set_quadrature_rules) are not walked here");
    int rc = source_ready(c, which ? "fh_estimate_H1_seminorm_error_squared" : "fh_estimate_L2_error_squared");
    if (rc) return rc;
    const int D = c->ei.d;
    if (sdim != 1 && (int)sdim != D) return c->fail(FH_BAD_ARGUMENT, "error estimate: solution dim must be 1 or the geometry dim");
    if (!uh_dev || !exact_dev || !out) return c->fail(FH_BAD_ARGUMENT, "error estimate: null argument");
    *out = 0.0;
    if (c->E == 0) return FH_OK;
    rc = reset_status(c);
    if (rc) return rc;
    KArgs a;
    fill_common(c, a);
    SourceArgs sa{};
    sa.N = c->ei.n;
    sa.NG = c->ei.ng;
    sa.phigeom = c->phigeom.p;
    const long long total = (long long)c->E * c->nq;
    const int grid = (int)std::min<long long>(2048, (total + 255) / 256);
    DevBuf<double> partial;
    HIP_TRY(c, partial.alloc(grid));
#define ERRK(DD, SS, WW) hipLaunchKernelGGL((k_error_squared<DD, SS, WW>), dim3(grid), dim3(256), 0, c->stream, a, sa, uh_dev, exact_dev, partial.p)
    if (D == 2 && sdim == 1) { if (which) ERRK(2, 1, 1); else ERRK(2, 1, 0); }
    else if (D == 2)         { if (which) ERRK(2, 2, 1); else ERRK(2, 2, 0); }
    else if (sdim == 1)      { if (which) ERRK(3, 1, 1); else ERRK(3, 1, 0); }
    else                     { if (which) ERRK(3, 3, 1); else ERRK(3, 3, 0); }
#undef ERRK
    HIP_TRY(c, hipGetLastError());
    rc = sum_partials(c, partial.p, grid, 1, out);
    if (rc) return rc;
    return which ? read_status(c, nullptr) : FH_OK;
}

static int error_squared_host(fh_ctx* c, int which, uint32_t sdim, const double* uh, const double* exact, double* out) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = source_ready(c, "error estimate");
    if (rc) return rc;
    if (!uh || !exact || !out) return c->fail(FH_BAD_ARGUMENT, "error estimate: null argument");
    const size_t n = (size_t)sdim * c->N, ne = (size_t)c->E * c->nq * sdim * (which ? c->ei.d : 1);
    DevBuf<double> du, de;
    HIP_TRY(c, du.alloc(n + 1));
    HIP_TRY(c, de.alloc(ne + 1));
    HIP_TRY(c, hipMemcpyAsync(du.p, uh, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(de.p, exact, sizeof(double) * ne, hipMemcpyHostToDevice, c->stream));
    return error_squared(c, which, sdim, du.p, de.p, out);
}

int fh_estimate_L2_error_squared(fh_ctx* c, uint32_t sdim, const double* u_h, const double* u_exact, double* out) {
    return c ? error_squared_host(c, 0, sdim, u_h, u_exact, out) : FH_BAD_ARGUMENT;
}
int fh_estimate_L2_error_squared_dev(fh_ctx* c, uint32_t sdim, const double* u_h_dev, const double* u_exact_dev, double* out) {
    return c ? error_squared(c, 0, sdim, u_h_dev, u_exact_dev, out) : FH_BAD_ARGUMENT;
}
int fh_estimate_H1_seminorm_error_squared(fh_ctx* c, uint32_t sdim, const double* u_h, const double* grad_exact, double* out) {
    return c ? error_squared_host(c, 1, sdim, u_h, grad_exact, out) : FH_BAD_ARGUMENT;
}
int fh_estimate_H1_seminorm_error_squared_dev(fh_ctx* c, uint32_t sdim, const double* u_h_dev, const double* grad_exact_dev, double* out) {
    return c ? error_squared(c, 1, sdim, u_h_dev, grad_exact_dev, out) : FH_BAD_ARGUMENT;
}

}  // extern "C"
