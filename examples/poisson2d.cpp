// C++ host driving the engine through the C ABI -- the shape of fenris's examples/poisson2d.rs
// (assemble_linear_system, examples/poisson2d.rs:33-60): Quad4 mesh of the unit square, 2x2 Gauss rule,
// Laplace operator, global stiffness matrix in CSR.  Build:  make -C examples   Run: ./examples/poisson2d [cells]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/fenris_hip.h"

#define CHECK(call)                                                                            \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != FH_OK) {                                                                    \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, fh_last_error(ctx));      \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

int main(int argc, char** argv) {
    const uint64_t cells = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 64;
    // create_unit_square_uniform_quad_mesh_2d(cells)  (src/mesh/procedural.rs:15-20)
    const double top_left[2] = {0.0, 1.0};
    uint64_t nv = 0, nc = 0;
    fh_quad_mesh_2d(1.0, 1, 1, cells, top_left, nullptr, nullptr, &nv, &nc);
    std::vector<double> vertices(2 * nv);
    std::vector<uint64_t> connectivity(4 * nc);
    fh_quad_mesh_2d(1.0, 1, 1, cells, top_left, vertices.data(), connectivity.data(), &nv, &nc);
    // quadrature::tensor::quadrilateral_gauss(2)
    std::vector<double> w(4), xi(8);
    fh_quadrilateral_gauss(2, w.data(), xi.data());

    fh_ctx* ctx = fh_create(0);
    if (!ctx) {
        std::fprintf(stderr, "no HIP device\n");
        return 2;
    }
    CHECK(fh_set_mesh(ctx, FH_QUAD4, vertices.data(), nv, connectivity.data(), nc));
    CHECK(fh_set_operator(ctx, FH_LAPLACE));
    CHECK(fh_set_quadrature_uniform(ctx, w.data(), xi.data(), 4, nullptr));
    CHECK(fh_set_u(ctx, nullptr));
    // CsrAssembler::assemble: pattern + values
    std::vector<uint64_t> row_offsets(nv + 1);
    uint64_t nnz = 0;
    CHECK(fh_pattern(ctx, row_offsets.data(), &nnz));
    std::vector<uint64_t> col_indices(nnz);
    CHECK(fh_pattern_cols(ctx, col_indices.data()));
    std::vector<double> values(nnz, 0.0);
    uint64_t failed = 0;
    CHECK(fh_assemble_matrix(ctx, values.data(), FH_SCATTER_GATHER, &failed));
    // constants are in the null space of the Laplace stiffness matrix: every row sums to zero
    double max_row_sum = 0.0, trace = 0.0;
    for (uint64_t r = 0; r < nv; ++r) {
        double s = 0.0;
        for (uint64_t k = row_offsets[r]; k < row_offsets[r + 1]; ++k) {
            s += values[k];
            if (col_indices[k] == r) trace += values[k];
        }
        max_row_sum = std::fmax(max_row_sum, std::fabs(s));
    }
    std::printf("Quad4 %llux%llu: %llu nodes, nnz %llu, trace %.12g, max |row sum| %.3e\n", (unsigned long long)cells,
                (unsigned long long)cells, (unsigned long long)nv, (unsigned long long)nnz, trace, max_row_sum);
    fh_destroy(ctx);
    return max_row_sum < 1e-12 ? 0 : 3;
}
