/*
 * fenris_oracle.h -- CPU restatement ("oracle") of the fenris global stiffness / residual
 * assembly path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product.  Only
 * tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load this library,
 * and there only as the checker / reported CPU baseline -- never as the thing measured or
 * shipped.  The product (fenris_amd/, include/) must not link, import or call it.
 *
 * The reference (InteractiveComputerGraphics/fenris, Rust) cannot be built in this image (no
 * cargo/rustc, no vendored crates, no network), so this file restates the algorithm statement
 * for statement in plain C.  Every function cites the reference file:line it follows (paths
 * relative to the reference checkout).
 *
 * PARITY PIN STATUS
 *   - index arrays (pattern, colouring, mesh generators): pinned bit-exactly by the reference's
 *     own known-answer tests and insta snapshots (tests/test_oracle_kat.py).
 *   - floating point values: pinned by the reference's analytic KATs (Lame conversion, material
 *     energies, Quad4 Laplace/mass element matrices), finite-difference consistency properties and
 *     the MMS error JSONs to the tolerances those tests state.  Dense 2x2/3x3 determinant/inverse
 *     and small mat-mul live in nalgebra 0.32.1 (un-vendored); their published formulae are
 *     restated here.  No reference test pins bit patterns of assembled values, so value parity is
 *     "pinned to 1e-12 relative; bit-level parity unpinned".
 *
 * Conventions (all as in the reference):
 *   - all matrices column-major (nalgebra); K_e is (s*n) x (s*n); dof = s*node + comp
 *     (src/assembly/global.rs:163-164)
 *   - vertices AoS [x,y(,z)] doubles; connectivity uint64_t[E*n] (usize) (src/mesh.rs:23-40)
 *   - compile with -ffp-contract=off: rustc never fuses a*b+c.
 */
#ifndef FENRIS_ORACLE_H
#define FENRIS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element kinds (node orders: SURVEY Appendix A.2) */
enum { FO_QUAD4 = 0, FO_HEX8 = 1, FO_TET4 = 2, FO_HEX27 = 3, FO_TRI3 = 4,
       /* quadratic, sub-parametric: tetrahedron.rs:92-246, quadrilateral.rs:150-330, triangle.rs:130-260 */
       FO_TET10 = 5, FO_QUAD9 = 6, FO_TRI6 = 7,
       /* 20-node serendipity hexahedron, hexahedron.rs:357-563 */
       FO_HEX20 = 8,
       /* cubic tetrahedron, tetrahedron.rs:248-470 */
       FO_TET20 = 9 };
/* operator kinds */
enum { FO_LAPLACE = 0, FO_LINEAR_ELASTIC = 1, FO_NEO_HOOKEAN = 2, FO_STVK = 3,
       /* ElementMassAssembler (src/assembly/local/mass.rs) with solution_dim 1 / geometry dim; q_params[2q] = density */
       FO_MASS_SCALAR = 4, FO_MASS_VECTOR = 5,
       /* a contraction given as DATA: C(a, b)[i][k] = sum_jl a[j] A[i][j][k][l] b[l], one d x d x d x d tensor per quadrature point, s = d.
        * What an EllipticContraction (src/assembly/operators.rs:146-189) whose coefficients do not depend on grad u computes; symmetric or not. */
       FO_TENSOR = 6 };
/* status codes */
enum { FO_OK = 0, FO_SINGULAR_JACOBIAN = 1, FO_BAD_ARGUMENT = 2, FO_COLUMN_NOT_FOUND = 4 };

int fo_element_num_nodes(int elem_kind);
int fo_element_dim(int elem_kind);
int fo_operator_solution_dim(int op_kind, int geom_dim);

/* ---- quadrature (fenris-quadrature/src/univariate.rs:66-118, tensor.rs:13-55, polyquad tables) */
int fo_gauss(int n, double* weights, double* points);
int fo_quadrilateral_gauss(int n, double* weights, double* points /* n*n*2 */);
int fo_hexahedron_gauss(int n, double* weights, double* points /* n*n*n*3 */);
/* returns number of points (or -1); strengths 1,2,3 tabulated (rules/polyquad/expanded/tet/{1-1,2-4,3-8}.txt) */
int fo_tetrahedron_rule(int strength, double* weights, double* points);
int fo_triangle_rule(int strength, double* weights, double* points);

/* ---- mesh generators (src/mesh/procedural.rs) ; outputs malloc'ed, release with fo_free */
void fo_free(void* p);
int fo_create_rectangular_uniform_quad_mesh_2d(double unit_length, uint64_t units_x, uint64_t units_y,
                                               uint64_t cells_per_unit, const double top_left[2],
                                               double** vertices, uint64_t* num_vertices,
                                               uint64_t** connectivity, uint64_t* num_cells);
int fo_create_rectangular_uniform_hex_mesh(double unit_length, uint64_t units_x, uint64_t units_y,
                                           uint64_t units_z, uint64_t cells_per_unit,
                                           double** vertices, uint64_t* num_vertices,
                                           uint64_t** connectivity, uint64_t* num_cells);
int fo_create_rectangular_uniform_tet_mesh(double unit_length, uint64_t units_x, uint64_t units_y,
                                           uint64_t units_z, uint64_t cells_per_unit,
                                           double** vertices, uint64_t* num_vertices,
                                           uint64_t** connectivity, uint64_t* num_cells);
/* Hex8 -> Hex27 (src/mesh_convert.rs:85-166,227-330) */
int fo_hex8_to_hex27(const double* vertices, uint64_t num_vertices, const uint64_t* hex8, uint64_t num_cells,
                     double** out_vertices, uint64_t* out_num_vertices, uint64_t** out_connectivity);

/* p-refinement Tet4 -> Tet10, Hex8 -> Hex20, Tri3 -> Tri6, Quad4 -> Quad9 (src/mesh_convert.rs:42-83, 168-217, 332-452);
 * outputs malloc'ed */
/* Tet20Mesh::from(&tet4) (src/mesh_convert.rs:658-775); outputs malloc'ed */
int fo_tet4_to_tet20(const double* vertices, uint64_t num_vertices, const uint64_t* tet4, uint64_t num_cells,
                     double** out_vertices, uint64_t* out_num_vertices, uint64_t** out_connectivity);
int fo_refine_to_quadratic(int from_kind, const double* vertices, uint64_t num_vertices, const uint64_t* connectivity,
                           uint64_t num_cells, double** out_vertices, uint64_t* out_num_vertices, uint64_t** out_connectivity);

/* ---- elements (src/element/ *.rs) */
/* reference gradients, d x n column-major (column per node) */
int fo_element_gradients(int elem_kind, const double* xi, double* grad);
int fo_element_basis(int elem_kind, const double* xi, double* phi);
/* J = X * G^T with X = d x n vertex matrix; elem_vertices AoS n*d */
int fo_element_reference_jacobian(int elem_kind, const double* elem_vertices, const double* xi, double* jac);

/* ---- materials (fenris-solid/src/materials.rs) */
void fo_lame_from_young_poisson(double young, double poisson, double* mu, double* lambda);
/* F is d x d column-major */
double fo_material_energy_density(int op_kind, int d, const double* F, double mu, double lambda);
void fo_material_stress_tensor(int op_kind, int d, const double* F, double mu, double lambda, double* P);
void fo_material_stress_contraction(int op_kind, int d, const double* F, const double* a, const double* b,
                                    double mu, double lambda, double* C);

/* ---- element assembler descriptor = ElementEllipticAssembler<Mesh, Op, UniformQuadratureTable>
 *      (src/assembly/local/elliptic.rs:152-158) */
typedef struct {
    int elem_kind;
    int op_kind;
    const double* vertices;   /* N x d AoS */
    uint64_t num_nodes;       /* N */
    const uint64_t* connectivity; /* E x n */
    uint64_t num_elements;    /* E */
    const double* u;          /* s*N (may be NULL => zeros) */
    const double* q_weights;  /* nq */
    const double* q_points;   /* nq x d */
    uint32_t nq;
    const double* q_params;   /* nq x 2 (mu, lambda) per point, or NULL for Laplace */
    /* CompactQuadratureTable (src/assembly/local/quadrature_table.rs:300-439) restricted to rules that share the
     * points and weights above and differ in their per-point data: element e uses rule_params[elem_to_rule[e]]
     * (num_rules x nq x 2).  NULL: the uniform table. */
    const uint64_t* elem_to_rule;
    const double* rule_params;
    uint64_t num_rules;
    /* FO_TENSOR only: nq x d^4 doubles, index ((i d + j) d + k) d + l; tensor_symmetric != 0: Symmetry::Symmetric (operators.rs:176-181:
     * only I <= J is filled, then clone_upper_to_lower), 0: Symmetry::NonSymmetric (every block filled, nothing mirrored) */
    const double* q_tensor;
    int tensor_symmetric;
} fo_assembler;

/* per-element kernels (src/assembly/local/elliptic.rs:361-439, 457-531, 551-605) */
int fo_assemble_element_matrix(const fo_assembler* a, uint64_t element, double* ke /* (s n)^2 col-major */);
int fo_assemble_element_vector(const fo_assembler* a, uint64_t element, double* fe /* s n */);
int fo_assemble_element_scalar(const fo_assembler* a, uint64_t element, double* energy);

/* ---- global (src/assembly/global.rs) */
/* generic (ragged) connectivity pattern, CsrAssembler::assemble_pattern global.rs:65-120.
 * elem_offsets has E+1 entries into elem_nodes.  Two-phase: pass col_indices=NULL to get nnz. */
int fo_assemble_pattern(uint64_t sdim, uint64_t num_nodes, uint64_t num_elements, const uint64_t* elem_offsets,
                        const uint64_t* elem_nodes, uint64_t* row_offsets /* sdim*N+1 */, uint64_t* col_indices,
                        uint64_t* nnz_out);
/* sequential_greedy_coloring fenris-paradis/src/coloring.rs:6-70.  color_offsets must hold E+2 entries
 * (worst case E colours); labels holds E entries (elements grouped by colour, order preserved). */
int fo_color_elements(uint64_t num_elements, const uint64_t* elem_offsets, const uint64_t* elem_nodes,
                      uint64_t* num_colors, uint64_t* color_offsets, uint64_t* labels);
/* CsrAssembler::assemble_into_csr global.rs:133-182 (serial, element order; accumulates) */
int fo_assemble_into_csr(const fo_assembler* a, const uint64_t* row_offsets, const uint64_t* col_indices,
                         double* values, uint64_t* failed_element);
/* CsrParAssembler::assemble_into_csr global.rs:314-376 (colour by colour, OpenMP threads) */
int fo_par_assemble_into_csr(const fo_assembler* a, uint64_t num_colors, const uint64_t* color_offsets,
                             const uint64_t* labels, const uint64_t* row_offsets, const uint64_t* col_indices,
                             double* values, int num_threads, uint64_t* failed_element);
/* VectorAssembler::assemble_vector_into global.rs:582-608 and coloured twin :643-685 */
int fo_assemble_vector_into(const fo_assembler* a, double* out, uint64_t* failed_element);
int fo_par_assemble_vector_into(const fo_assembler* a, uint64_t num_colors, const uint64_t* color_offsets,
                                const uint64_t* labels, double* out, int num_threads, uint64_t* failed_element);
/* ElementSourceAssembler (src/assembly/local/source.rs:159-278) through VectorAssembler; see the .c file for the
 * two source kinds (uniform density * g, or values sampled at the physical quadrature points) */
int fo_assemble_element_source_vector(const fo_assembler* a, uint64_t element, int s, const double* g,
                                      const double* values, double* fe /* s n */);
int fo_assemble_source_vector_into(const fo_assembler* a, int s, const double* g, const double* values, double* out);
int fo_physical_quadrature_points(const fo_assembler* a, double* x_out /* E nq d */);
/* assemble_scalar global.rs:697-711 */
int fo_assemble_scalar(const fo_assembler* a, double* out, uint64_t* failed_element);
/* apply_homogeneous_dirichlet_bc_csr global.rs:379-451 / rhs :479-495 */
int fo_apply_homogeneous_dirichlet_bc_csr(uint64_t num_rows, const uint64_t* row_offsets, const uint64_t* col_indices,
                                          double* values, const uint64_t* nodes, uint64_t num_bc_nodes,
                                          uint64_t solution_dim);
/* ConjugateGradient::solve_with_guess (fenris-sparse/src/cg.rs:366-478) on a CSR matrix; see the .c file */
int fo_cg_solve(uint64_t n, const uint64_t* row_offsets, const uint64_t* col_indices, const double* values, const double* b,
                double* x, int jacobi, double tol, uint64_t max_iter, uint64_t* num_iterations);
/* estimate_L2_error_squared (which = 0) / estimate_H1_seminorm_error_squared (which = 1), src/error.rs:287-372 */
int fo_estimate_error_squared(const fo_assembler* a, int which, int s, const double* u_h, const double* exact, double* out);
/* cuthill_mckee / reorder_mesh_par (src/mesh/reorder.rs:54-95, 171-239); perm[target] = source */
int fo_cuthill_mckee(uint64_t n, const uint64_t* row_offsets, const uint64_t* col_indices, uint64_t* perm);
int fo_reorder_mesh(uint64_t num_vertices, uint64_t nodes_per_element, const uint64_t* connectivity, uint64_t num_elements,
                    uint64_t* vertex_perm, uint64_t* connectivity_perm);
int fo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
