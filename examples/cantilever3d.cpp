// C++ host driving the whole chain through the C ABI, no Python: a Hex8 cantilever under its own weight.
//   mesh generator -> two contexts on the same mesh (stiffness operator, gravity source) -> CSR pattern ->
//   owner-computes stiffness assembly -> GravitySource load vector -> homogeneous Dirichlet conditions on the x = 0
//   face -> Jacobi-preconditioned CG.  Everything between fh_assemble_* and the solve stays on the device.
// Mirrors what a fenris application does with CsrAssembler / ElementSourceAssembler / apply_homogeneous_dirichlet_bc_*
// / ConjugateGradient (src/assembly/global.rs, src/assembly/local/source.rs, fenris-sparse/src/cg.rs).
// Build:  make -C examples    Run: ./examples/cantilever3d [cells_per_unit]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/fenris_hip.h"

#define CHECK(ctx, call)                                                                       \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != FH_OK) {                                                                    \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, fh_last_error(ctx));      \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)
#define HIP(call)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                    \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

int main(int argc, char** argv) {
    const uint64_t cpu = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 8;
    // beam 4 x 1 x 1: create_rectangular_uniform_hex_mesh(1.0, 4, 1, 1, cells_per_unit)  (procedural.rs:216-277)
    uint64_t nv = 0, nc = 0;
    fh_hex_mesh(1.0, 4, 1, 1, cpu, nullptr, nullptr, &nv, &nc);
    std::vector<double> vertices(3 * nv);
    std::vector<uint64_t> connectivity(8 * nc);
    fh_hex_mesh(1.0, 4, 1, 1, cpu, vertices.data(), connectivity.data(), &nv, &nc);
    std::vector<double> w(8), xi(24);
    fh_hexahedron_gauss(2, w.data(), xi.data());
    double mu, lambda;
    fh_lame_from_young_poisson(1e7, 0.3, &mu, &lambda);
    std::vector<double> lame(16), density(16);
    for (int q = 0; q < 8; ++q) { lame[2 * q] = mu; lame[2 * q + 1] = lambda; density[2 * q] = 1000.0; density[2 * q + 1] = 0.0; }

    fh_ctx* k_ctx = fh_create(0);
    fh_ctx* f_ctx = fh_create(0);
    if (!k_ctx || !f_ctx) { std::fprintf(stderr, "no HIP device\n"); return 2; }
    CHECK(k_ctx, fh_set_mesh(k_ctx, FH_HEX8, vertices.data(), nv, connectivity.data(), nc));
    CHECK(k_ctx, fh_set_operator(k_ctx, FH_LINEAR_ELASTIC));
    CHECK(k_ctx, fh_set_quadrature_uniform(k_ctx, w.data(), xi.data(), 8, lame.data()));
    CHECK(k_ctx, fh_set_u(k_ctx, nullptr));
    CHECK(f_ctx, fh_set_mesh(f_ctx, FH_HEX8, vertices.data(), nv, connectivity.data(), nc));
    CHECK(f_ctx, fh_set_quadrature_uniform(f_ctx, w.data(), xi.data(), 8, density.data()));

    const uint64_t n = 3 * nv;
    std::vector<uint64_t> row_offsets(n + 1);
    uint64_t nnz = 0;
    CHECK(k_ctx, fh_pattern(k_ctx, row_offsets.data(), &nnz));
    double *values = nullptr, *rhs = nullptr, *u = nullptr;
    HIP(hipMalloc(&values, sizeof(double) * nnz));
    HIP(hipMalloc(&rhs, sizeof(double) * n));
    HIP(hipMalloc(&u, sizeof(double) * n));
    HIP(hipMemset(rhs, 0, sizeof(double) * n));
    HIP(hipMemset(u, 0, sizeof(double) * n));
    uint64_t failed = 0;
    CHECK(k_ctx, fh_assemble_matrix_dev(k_ctx, values, FH_SCATTER_GATHER | FH_ASSEMBLE_OVERWRITE, &failed));
    const double g[3] = {0.0, 0.0, -9.81};
    CHECK(f_ctx, fh_assemble_source_vector_dev(f_ctx, 3, g, nullptr, rhs));
    std::vector<uint64_t> clamped;
    for (uint64_t i = 0; i < nv; ++i)
        if (vertices[3 * i] < 1e-12) clamped.push_back(i);
    CHECK(k_ctx, fh_apply_dirichlet_csr_dev(k_ctx, values, clamped.data(), clamped.size()));
    CHECK(k_ctx, fh_apply_dirichlet_rhs_dev(k_ctx, rhs, clamped.data(), clamped.size()));
    uint64_t iterations = 0;
    CHECK(k_ctx, fh_cg_solve_dev(k_ctx, values, rhs, u, FH_PRECOND_JACOBI, 1e-8, 20000, &iterations));

    std::vector<double> uh(n), fh(n);
    HIP(hipMemcpy(uh.data(), u, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(fh.data(), rhs, sizeof(double) * n, hipMemcpyDeviceToHost));
    double tip = 0.0, total_load = 0.0, clamped_motion = 0.0;
    for (uint64_t i = 0; i < nv; ++i) {
        if (vertices[3 * i] > 4.0 - 1e-12) tip = std::fmin(tip, uh[3 * i + 2]);
        total_load += fh[3 * i + 2];
    }
    for (uint64_t i : clamped) clamped_motion = std::fmax(clamped_motion, std::fabs(uh[3 * i]) + std::fabs(uh[3 * i + 1]) + std::fabs(uh[3 * i + 2]));
    // load check: the free nodes carry rho g V minus the share of the clamped face; Euler-Bernoulli tip deflection
    // q L^4 / (8 E I) with q = rho g A, I = 1/12 as an order-of-magnitude reference
    const double eb = 1000.0 * 9.81 * 1.0 * 256.0 / (8.0 * 1e7 / 12.0);
    std::printf("Hex8 cantilever %llux%llux%llu: %llu dofs, nnz %llu, CG iterations %llu (%s), tip deflection %.6e "
                "(Euler-Bernoulli %.3e), sum of nodal loads %.6g, clamped face moves %.1e\n",
                (unsigned long long)(4 * cpu), (unsigned long long)cpu, (unsigned long long)cpu, (unsigned long long)n,
                (unsigned long long)nnz, (unsigned long long)iterations, fh_last_kernel_name(k_ctx), tip, -eb, total_load,
                clamped_motion);
    (void)hipFree(values); (void)hipFree(rhs); (void)hipFree(u);
    fh_destroy(k_ctx);
    fh_destroy(f_ctx);
    const bool ok = iterations > 0 && clamped_motion == 0.0 && tip < 0.0 && std::fabs(tip / -eb) > 0.5 && std::fabs(tip / -eb) < 2.0;
    return ok ? 0 : 3;
}
