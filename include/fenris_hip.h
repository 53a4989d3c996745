/*
 * fenris_hip.h -- C ABI of the MI355X-native FEM assembly engine (libfenris_hip.so).
 *
 * This is the drop-in boundary for fenris's global stiffness / residual assembly path: every entry
 * point names the reference interface it replaces (paths relative to the fenris checkout).  The
 * reference's plugin point is the Element*Assembler trait family (src/assembly/local.rs:18-149)
 * consumed by CsrAssembler / CsrParAssembler / VectorAssembler (src/assembly/global.rs); arbitrary
 * Rust element assemblers cannot run on a GPU, so the engine implements the closed family the
 * reference ships -- ElementEllipticAssembler<Mesh<C>, Op, UniformQuadratureTable>
 * (src/assembly/local/elliptic.rs:152-340) -- selected through plain descriptors.
 *
 * Conventions
 *   - plain pointers and sizes only; `usize` of the reference is uint64_t; all reals are f64.
 *   - vertices are AoS [x,y(,z)] (Vec<OPoint<f64,D>>, src/mesh.rs:23-40); connectivity is E x n
 *     uint64_t (Vec<[usize; n]>, src/connectivity.rs:606-607) -- both are zero-copy views of the
 *     reference's own storage.
 *   - dof numbering: s*node + component (src/assembly/global.rs:163-164); CSR as in nalgebra-sparse
 *     (row_offsets[R+1], col_indices[nnz] ascending per row, values[nnz]).
 *   - pointers are HOST pointers unless the function name ends in _dev (then: device pointers valid on
 *     the context's device; the call is enqueued on the context's stream and returns after enqueueing
 *     unless it has a host out-parameter, in which case it synchronises the stream).
 *   - every function returns an int32 status; nothing aborts.  fh_last_error() gives the message.
 *   - a context is bound to one device and one host thread at a time (thread-compatible, like
 *     CsrAssembler which is !Sync, src/assembly/global.rs:27-31).
 */
#ifndef FENRIS_HIP_H
#define FENRIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FH_ABI_VERSION 1

/* Tuning and diagnostics: environment variables named FENRIS_HIP_* (scripts/README.md) are read ONCE, by fh_create, into the
 * context; no later call reads the environment.  They change which kernel variant runs or print diagnostics, never results
 * beyond rounding -- except FENRIS_HIP_ABLATE / FENRIS_HIP_TRACE, which select instrumented instantiations for profiling
 * (FENRIS_HIP_ABLATE switches work off and produces wrong values by design). */

/* status codes.  FH_SINGULAR_JACOBIAN is the reference's only runtime error on the path:
 * eyre!("Singular element Jacobian encountered"), src/assembly/local/elliptic.rs:401-404. */
enum {
    FH_OK = 0,
    FH_SINGULAR_JACOBIAN = 1,
    FH_BAD_ARGUMENT = 2,
    FH_HIP_ERROR = 3,         /* a HIP or RCCL call failed; fh_last_error has the text */
    FH_OUT_OF_MEMORY = 4,     /* a buffer the call needs does not fit: device memory, or a cap such as FENRIS_HIP_TWO_PASS_MAX_GB (round 6; the value was reserved) */
    FH_INVALID_STATE = 5,     /* e.g. assemble before pattern, operator/element dimension mismatch */
    FH_UNSUPPORTED = 6,
    /* SolveErrorKind of the conjugate-gradient solver (fenris-sparse/src/cg.rs:277-286) */
    FH_CG_MAX_ITERATIONS = 7,
    FH_CG_INDEFINITE_OPERATOR = 8,
    FH_CG_INDEFINITE_PRECONDITIONER = 9
};

/* element kinds: Quad4d2Element (src/element/quadrilateral.rs:70-142), Hex8Element
 * (src/element/hexahedron.rs:34-117), Tet4Element (src/element/tetrahedron.rs:543-608),
 * Hex27Element (hexahedron.rs:157-335), Tri3d2Element (src/element/triangle.rs:63-110) */
enum { FH_QUAD4 = 0, FH_HEX8 = 1, FH_TET4 = 2, FH_HEX27 = 3, FH_TRI3 = 4,
       /* quadratic elements, sub-parametric like Hex27 -- the geometry map is the embedded linear element's:
        * Tet10Element (src/element/tetrahedron.rs:92-246), Quad9d2Element (quadrilateral.rs:150-330),
        * Tri6d2Element (triangle.rs:130-260) */
       FH_TET10 = 5, FH_QUAD9 = 6, FH_TRI6 = 7,
       /* Hex20Element, 20-node serendipity (src/element/hexahedron.rs:357-563), geometry from the embedded Hex8 */
       FH_HEX20 = 8,
       /* Tet20Element, cubic (src/element/tetrahedron.rs:248-470), geometry from the embedded Tet4 */
       FH_TET20 = 9 };

/* operator kinds: LaplaceOperator (src/assembly/operators/laplace.rs), MaterialEllipticOperator over
 * LinearElasticMaterial / NeoHookeanMaterial / StVKMaterial (fenris-solid/src/lib.rs:412-508,
 * fenris-solid/src/materials.rs:83-123, 236-353, 392-469) */
enum { FH_LAPLACE = 0, FH_LINEAR_ELASTIC = 1, FH_NEO_HOOKEAN = 2, FH_STVK = 3,
       /* ElementMassAssembler::with_solution_dim(1 | D) (src/assembly/local/mass.rs:48-286): M_IJ = I_s sum_q w |det J|
        * rho phi_I phi_J; the per-point parameter pair carries Density(rho) in its first slot.  Matrix only. */
       FH_MASS_SCALAR = 4, FH_MASS_VECTOR = 5,
       /* An elliptic operator given as DATA instead of code (round 6): the contraction of an EllipticContraction (src/assembly/operators.rs:146-189)
        * whose coefficients do not depend on grad u,
        *     C(a, b)[i][k] = sum_{j, l} a[j] A[i][j][k][l] b[l],       s = d,
        * with one tensor A per quadrature point (fh_set_operator_tensor).  Covers every LINEAR elliptic operator -- anisotropic elasticity,
        * a linearisation frozen at some state, operators that are not symmetric -- without a closure crossing the boundary; a caller
        * with a nonlinear operator evaluates its tangent at the quadrature data it owns.  Stiffness matrix only (all scatter modes). */
       FH_TENSOR = 6 };

/* how K_e contributions reach the CSR values (flags argument of fh_assemble_matrix*):
 *   FH_SCATTER_ATOMIC  : element-parallel, fp64 atomic adds (replaces the rayon colour loop)
 *   FH_SCATTER_COLORED : one launch per colour, plain read-modify-write -- CsrParAssembler semantics
 *                        (src/assembly/global.rs:314-376); needs fh_color() or fh_set_colors()
 *   FH_SCATTER_GATHER  : owner-computes: each CSR row block is produced by one workgroup from all
 *                        elements adjacent to its node and written once, coalesced; no atomics on HBM
 * OR-in FH_ASSEMBLE_OVERWRITE to store K instead of accumulating into the existing values
 * (= CsrAssembler::assemble, global.rs:124-131, without the explicit zero fill). */
enum { FH_SCATTER_ATOMIC = 0, FH_SCATTER_COLORED = 1, FH_SCATTER_GATHER = 2, FH_SCATTER_MASK = 0xff };
enum { FH_ASSEMBLE_OVERWRITE = 0x100 };
/* OR-in FH_ASSEMBLE_REPRODUCIBLE (with FH_SCATTER_GATHER) to get the same bits from run to run and from launch geometry to launch geometry, like the
 * reference's coloured loop (global.rs:322-373: every entry is a sum in a fixed order).  The row-owner kernels (affine and general Hex8 with the
 * eight-point rule, Tet4) and the two-pass form (Hex27, NeoHookean, StVK) already are; the configurations whose one-pass kernel accumulates with
 * LDS atomics in hardware order (Quad4 / Tri3, Hex8 with other rules or per-point parameters) take the two-pass form instead -- slower, and the
 * dense element matrices need E (s n)^2 doubles (9 E n (n + 1) / 2 for 3 x 3 blocks on the 3D elements: node-block triangles); FH_UNSUPPORTED under a row range.  FH_SCATTER_COLORED is reproducible as it is;
 * FH_SCATTER_ATOMIC never is (FH_BAD_ARGUMENT with this flag). */
enum { FH_ASSEMBLE_REPRODUCIBLE = 0x200 };

typedef struct fh_ctx fh_ctx;

/* ---- context ------------------------------------------------------------------------------- */
fh_ctx* fh_create(int device_id);               /* NULL if the device cannot be initialised */
void fh_destroy(fh_ctx*);
const char* fh_last_error(const fh_ctx*);
int fh_abi_version(void);
/* hipStream_t to launch on (NULL = default stream).  Not owned. */
int fh_set_stream(fh_ctx*, void* hip_stream);
int fh_synchronize(fh_ctx*);

/* ---- inputs -------------------------------------------------------------------------------- */
/* Mesh<f64, D, C>: replaces passing &Mesh to ElementEllipticAssemblerBuilder::with_finite_element_space
 * (src/assembly/local/elliptic.rs:86-97).  Data is copied to the device (connectivity narrowed to i32;
 * num_vertices must be < 2^31).  Invalidates pattern, colours and u. */
int fh_set_mesh(fh_ctx*, int elem_kind, const double* vertices, uint64_t num_vertices,
                const uint64_t* connectivity, uint64_t num_elements);
int fh_set_mesh_dev(fh_ctx*, int elem_kind, const double* vertices_dev, uint64_t num_vertices,
                    const uint64_t* connectivity_dev, uint64_t num_elements);
/* only the vertex coordinates change (e.g. moving mesh); pattern stays valid */
int fh_update_vertices(fh_ctx*, const double* vertices);
/* Generic ElementConnectivityAssembler with ragged element node lists (src/assembly/local.rs:18-47),
 * e.g. the mock connectivities of tests/unit_tests/assembly/global.rs:70-142.  Only fh_pattern*,
 * fh_color work on such a context.  elem_offsets has num_elements+1 entries. */
int fh_set_connectivity_ragged(fh_ctx*, uint64_t solution_dim, uint64_t num_nodes, const uint64_t* elem_offsets,
                               const uint64_t* elem_nodes, uint64_t num_elements);
/* Restrict the NUMERIC assembly (matrix, vector, scalar) to the elements with mask[e] != 0 while the sparsity
 * pattern keeps coming from all elements.  No reference counterpart (fenris is single-process): this is how a
 * mesh partition assembles its own elements into rows that carry the global pattern (own + halo elements),
 * see fenris_amd/distributed.py.  mask has num_elements bytes; NULL removes the mask. */
int fh_set_active_elements(fh_ctx*, const uint8_t* mask);
/* Restrict FH_SCATTER_GATHER assembly to the rows of the nodes [node_begin, node_end): only those CSR rows are
 * produced (the others are left untouched).  Together with fh_assemble_matrix_rows_dev (below) one context assembles
 * the matrix in two launches over complementary ranges -- the multi-GPU path launches the rows of a partition interface
 * first and sends them while the rest is computed (fenris_amd/distributed.py).  (0, num_nodes) restores the default. */
int fh_set_row_range(fh_ctx*, uint64_t node_begin, uint64_t node_end);
/* Operator: replaces .with_operator(&op) (elliptic.rs:99-108).  Solution dim s = 1 for Laplace, D else. */
int fh_set_operator(fh_ctx*, int op_kind);
/* The coefficient tensors of FH_TENSOR: nq x d^4 doubles, index ((i d + j) d + k) d + l, nq = the points of the quadrature table in use (set the
 * table first; a later fh_set_quadrature_* with another point count invalidates them).  symmetric != 0: the caller asserts
 * A[i][j][k][l] == A[k][l][i][j], i.e. Symmetry::Symmetric -- only the blocks I <= J are formed and the rest mirrored exactly like for the
 * built-in operators (operators.rs:176-181, util.rs:38-51); 0: Symmetry::NonSymmetric -- every block of every row is formed (operators.rs:180),
 * nothing is mirrored, and the assembled matrix is not symmetric. */
int fh_set_operator_tensor(fh_ctx*, const double* tensors, uint32_t nq, int symmetric);
/* UniformQuadratureTable::from_points_and_weights(points, weights).with_data / with_uniform_data
 * (src/assembly/local/quadrature_table.rs:213-298).  params: nq x 2 doubles (LameParameters{mu,lambda}
 * per point, fenris-solid/src/materials.rs:8-12) or NULL for operators without parameters. */
int fh_set_quadrature_uniform(fh_ctx*, const double* weights, const double* points, uint32_t nq,
                              const double* params);
/* The same table with the per-point data where and how the caller keeps it (`data: Vec<Parameters>` of
 * UniformQuadratureTable, quadrature_table.rs:213-298): `stride` bytes from one point's record to the next (a multiple of
 * 8), `kind` says what a record starts with -- FH_DATA_LAME: LameParameters {mu, lambda} (fenris-solid/src/materials.rs:8-12),
 * FH_DATA_DENSITY: Density(rho) of the mass / gravity assemblers (fenris-solid/src/gravity_source.rs), FH_DATA_NONE: `()`. */
enum { FH_DATA_NONE = 0, FH_DATA_LAME = 1, FH_DATA_DENSITY = 2 };
int fh_set_quadrature_uniform_data(fh_ctx*, const double* weights, const double* points, uint32_t nq, const void* data,
                                   uint32_t stride, int kind);
/* CompactQuadratureTable::from_quadrature_rules_and_map (src/assembly/local/quadrature_table.rs:300-439) for rules
 * that share points and weights and differ in their per-point data -- piecewise material parameters: element e
 * uses rule_params[elem_to_rule[e]] (num_rules x nq x 2, same pair layout as the uniform table).  A rule index out
 * of bounds is FH_BAD_ARGUMENT (the reference panics).  Rules with different point sets: fh_set_quadrature_rules below.
 * Rules whose data are the same at every point (one
 * LameParameters pair per element: the multi-material case) keep the fastest LinearElastic stiffness kernel, which then
 * reads the pair per element; rules that vary over their points take the per-point-coefficient kernels. */
int fh_set_quadrature_compact(fh_ctx*, const double* weights, const double* points, uint32_t nq, uint64_t num_rules,
                              const double* rule_params, const uint64_t* elem_to_rule);
/* Rule-set tables: GeneralQuadratureTable (one rule per element, src/assembly/local/quadrature_table.rs:57-210) and
 * CompactQuadratureTable::from_quadrature_rules_and_map with rules of DIFFERENT point sets (:300-439).  Rule r holds the
 * points [rule_offsets[r], rule_offsets[r + 1]) of `weights` (total), `points` (total x d) and `params` (total x 2, the pair
 * layout of the uniform table, or NULL for operators without parameters); element e uses rule elem_to_rule[e]
 * (NULL: rule e, then num_rules must equal the number of elements).  A rule index out of bounds or an empty rule is
 * FH_BAD_ARGUMENT (the reference panics: check_rules_consistency, quadrature_table.rs:366-372).
 * The engine groups the rules by (points, weights): a group runs as one uniform / compact device table over its elements,
 * and fh_assemble_matrix*, fh_assemble_vector*, fh_assemble_scalar and fh_assemble_element_matrices* walk the groups
 * inside the library, accumulating -- a table whose rules share their points costs one pass, E different point sets cost
 * E passes.  The source-vector, physical-point and error-estimate entry points answer FH_UNSUPPORTED while such a table
 * is set.  Replaced by the next fh_set_quadrature_* call; dropped by fh_set_mesh*. */
int fh_set_quadrature_rules(fh_ctx*, uint64_t num_rules, const uint64_t* rule_offsets, const double* weights,
                            const double* points, const double* params, const uint64_t* elem_to_rule);
/* number of (points, weights) groups of the rule-set table in use (0: none set) = passes per assembly */
int fh_quadrature_rule_groups(const fh_ctx*, uint64_t* num_groups);
/* Affine-element fast path of FH_SCATTER_GATHER (Hex8; Laplace / LinearElastic with uniform parameters).  On an element
 * whose geometry map is affine the Jacobian of elliptic.rs:399 is the same at every quadrature point, so
 * K_ab = |det J| C(J^-T Ghat_ab J^-1) with Ghat_ab = sum_q w_q ghat_a ghat_b^T depending on the rule only -- the engine
 * detects such elements from the vertex coordinates and runs the node blocks whose elements all qualify on a kernel
 * without a quadrature loop; every other block keeps the general kernels.  An element qualifies when the four mixed
 * coefficients of its trilinear map are at most rel_tol times its shortest edge-direction coefficient.  Results change
 * by O(rel_tol) relative at most (exactly affine elements -- every generated box mesh -- agree to rounding).
 * Default 2^-46 (1.4e-14); 0 switches the path off.  No reference counterpart (the reference has one code path).
 * Only the stiffness fast path follows a loosened tolerance: the residual / energy kernels take the all-affine shortcut of a mesh
 * only at the default tolerance (or tighter) and use the exact geometry otherwise. */
int fh_set_affine_tolerance(fh_ctx*, double rel_tol);
/* how the last FH_SCATTER_GATHER assembly was split: elements found affine, node blocks on the affine kernel, node blocks on
 * the general kernels (any pointer may be NULL; zeros before the first assembly) */
int fh_affine_stats(const fh_ctx*, uint64_t* affine_elements, uint64_t* affine_blocks, uint64_t* general_blocks);
/* .with_u(&u) (elliptic.rs:123-137); u has s*N entries; NULL = zeros */
int fh_set_u(fh_ctx*, const double* u);
int fh_set_u_dev(fh_ctx*, const double* u_dev);

/* ---- queries (ElementConnectivityAssembler, src/assembly/local.rs:18-27) ---------------------- */
uint64_t fh_solution_dim(const fh_ctx*);
uint64_t fh_num_elements(const fh_ctx*);
uint64_t fh_num_nodes(const fh_ctx*);
uint64_t fh_num_rows(const fh_ctx*);   /* s*N */
uint64_t fh_nnz(const fh_ctx*);        /* 0 before fh_pattern */

/* ---- sparsity pattern: CsrAssembler::assemble_pattern / CsrParAssembler::assemble_pattern
 *      (src/assembly/global.rs:65-120, 206-297); bit-identical output ---------------------------- */
/* builds the pattern on the device; row_offsets (R+1 entries) may be NULL; nnz_out may be NULL */
int fh_pattern(fh_ctx*, uint64_t* row_offsets, uint64_t* nnz_out);
int fh_pattern_cols(fh_ctx*, uint64_t* col_indices /* nnz */);
/* device outputs; either may be NULL.  Requires a prior fh_pattern(). */
int fh_pattern_dev(fh_ctx*, uint64_t* row_offsets_dev, uint64_t* col_indices_dev);

/* ---- colouring: color_nodes + sequential_greedy_coloring (src/assembly/global.rs:540-551,
 *      fenris-paradis/src/coloring.rs:6-70); reference-identical ----------------------------------- */
/* color_offsets: capacity num_elements+2 (or NULL); labels: num_elements entries (or NULL):
 * elements of colour c are labels[color_offsets[c] .. color_offsets[c+1]) in ascending order */
int fh_color(fh_ctx*, uint64_t* num_colors, uint64_t* color_offsets, uint64_t* labels);
/* the same outputs, computed ON the device: Luby-style rounds (propose the smallest colour no finished neighbour holds, keep it unless
 * a neighbour of smaller hashed priority proposed the same), deterministic.  A valid colouring -- no two elements of a colour share a
 * node, which is all CsrParAssembler / DisjointSubsets require -- but generally not the sequential greedy one of fh_color (more colours
 * are possible).  Fixed-size connectivity only. */
int fh_color_parallel(fh_ctx*, uint64_t* num_colors, uint64_t* color_offsets, uint64_t* labels);
/* reuse a colouring computed elsewhere (colours are serialisable in the reference, paradis lib.rs:170) */
int fh_set_colors(fh_ctx*, uint64_t num_colors, const uint64_t* color_offsets, const uint64_t* labels);

/* ---- numeric assembly ------------------------------------------------------------------------ */
/* CsrAssembler::assemble_into_csr / CsrParAssembler::assemble_into_csr (global.rs:133-182, 314-376):
 * values (nnz doubles, layout of the fh_pattern CSR) are ACCUMULATED into unless FH_ASSEMBLE_OVERWRITE.
 * On FH_SINGULAR_JACOBIAN *failed_element is the lowest failing element index (may be NULL) and the
 * values are unspecified (the reference aborts at the first failing element). */
int fh_assemble_matrix(fh_ctx*, double* values, int flags, uint64_t* failed_element);
int fh_assemble_matrix_dev(fh_ctx*, double* values_dev, int flags, uint64_t* failed_element);
/* same, but only enqueues; check the status later with fh_poll_status (no host sync; for timing loops).  The FIRST calls after a change of
 * the mesh, the pattern, the mask or the operator do block the host: the first builds the owner-computes tables (tens of milliseconds), and
 * on general Hex8 meshes the second runs the lane tuner of k_hex8_rows in front of its launch (a device synchronisation and ~20 ms of host
 * work, once; FENRIS_HIP_TUNE_AFTER moves it, FENRIS_HIP_NO_LANE_TUNING removes it).  fh_time_assembly_dev and fh_tune_placement_dev run both
 * before their timed assemblies. */
int fh_assemble_matrix_async_dev(fh_ctx*, double* values_dev, int flags);
int fh_poll_status(fh_ctx*, uint64_t* failed_element);
/* Placement of the streamed buffers.  On MI355X the time of the owner-computes kernels follows how the large buffers they stream
 * through happen to be backed by device memory: the same context and arguments run at one of several levels up to 10 % apart, for
 * the life of an allocation, and no HIP call chooses the backing.  fh_time_assembly_dev times `reps` assemblies (after one untimed)
 * with events on the context's stream; a caller uses it to keep the better of several allocations of its `values`.
 * fh_tune_placement_dev does the same for the library's own large buffer (the element records of the affine-element kernel): up to
 * `tries` re-allocations, each timed with three assemblies, the fastest kept.  BOTH need FH_ASSEMBLE_OVERWRITE (the timed / trial
 * assemblies are real ones and write `values`; FH_BAD_ARGUMENT otherwise).  No reference counterpart. */
/* A tuning switch of this context (a FENRIS_HIP_* name as fh_create reads them from the environment): set, or removed with value ==
 * NULL.  Launch-variant switches act at the next call.  For comparing variants inside ONE context on the same buffers. */
int fh_set_option(fh_ctx*, const char* name, const char* value);
/* Device memory through the virtual-memory API with an explicit physical chunk size (hipMemAddressReserve / hipMemCreate / hipMemMap): `bytes`
 * on `device` from chunks of `chunk_bytes` (rounded up to the allocation granularity, reported in *granularity_out; 0 = one chunk).  For
 * experiments on how a large `values` array is backed (profiles/r05_vmm_experiment.txt); free with fh_vmm_free.  No reference counterpart. */
int fh_vmm_alloc(int device, uint64_t bytes, uint64_t chunk_bytes, void** out, uint64_t* granularity_out);
int fh_vmm_free(void* ptr);
/* Host staging memory of the set-up stages comes from a process-wide pool that is never returned to the operating system while in use (unmapping
 * memory a device copy has touched suspends the process's GPU queues for tens of milliseconds, profiles/r05_setup.txt): at most 1 GiB is retained.
 * fh_host_pool_trim frees what the pool holds (call it when no latency-critical launch is near); returns the bytes freed. */
uint64_t fh_host_pool_trim(void);
int fh_time_assembly_dev(fh_ctx*, double* values_dev, int flags, int reps, double* ms_per_assembly);
int fh_tune_placement_dev(fh_ctx*, double* values_dev, int flags, int tries, double* ms_before, double* ms_after);
/* The CSR rows of the nodes [node_begin, node_end) only (FH_SCATTER_GATHER), whatever the context's own row range is: the
 * context keeps a second set of owner-computes tables for this range next to its own (mesh, pattern, quadrature and operator
 * are shared, nothing is duplicated), built on first use and rebuilt when the range or the context's configuration changes.
 * A context with fh_set_row_range(split, N) plus this call on [0, split) produces the matrix in two launches.  Not with
 * rule-set tables (fh_set_quadrature_rules).  The _async form only enqueues; fh_poll_status reports its errors too. */
int fh_assemble_matrix_rows_dev(fh_ctx*, double* values_dev, int flags, uint64_t node_begin, uint64_t node_end,
                                uint64_t* failed_element);
int fh_assemble_matrix_rows_async_dev(fh_ctx*, double* values_dev, int flags, uint64_t node_begin, uint64_t node_end);
/* VectorAssembler::assemble_vector_into / VectorParAssembler (global.rs:582-608, 643-685) with
 * assemble_element_elliptic_vector (elliptic.rs:457-531): out (s*N) is accumulated into. */
int fh_assemble_vector(fh_ctx*, double* out, uint64_t* failed_element);
int fh_assemble_vector_dev(fh_ctx*, double* out_dev, uint64_t* failed_element);
/* the same, only enqueued on the context's stream: a singular element is reported by the next fh_poll_status (like
 * fh_assemble_matrix_async_dev) -- also over a rule-set table (one launch per rule group; the status is reset once in front of them).
 * The FIRST assembly of a context still builds the pattern and the adjacency, which synchronises; later calls only enqueue. */
int fh_assemble_vector_async_dev(fh_ctx*, double* out_dev);
/* assemble_scalar (global.rs:697-711) with compute_element_elliptic_energy (elliptic.rs:551-605) */
int fh_assemble_scalar(fh_ctx*, double* out, uint64_t* failed_element);
/* ElementSourceAssembler through VectorAssembler (src/assembly/local/source.rs:159-278, global.rs:582-608):
 *   out[s node + c] += sum_e sum_q w |det J| f_c(e, q) phi_node(xi_q)
 * independent of fh_set_operator; solution_dim is 1 or the geometry dimension.  The reference's SourceFunction is
 * arbitrary code; the closed family behind this ABI:
 *   values == NULL: f(e, q) = density_q * g   -- GravitySource (fenris-solid/src/gravity_source.rs:57-65); density_q
 *                   is the first parameter of the quadrature table (Density<T>), g has solution_dim entries (host)
 *   values != NULL: f(e, q) = values[(e nq + q) solution_dim ..] -- any source, sampled by the caller at the physical
 *                   points returned by fh_physical_quadrature_points (x = map_reference_coords(xi_q), source.rs:263)
 * Only |det J| enters (source.rs:276): no singular-Jacobian error on this path. */
int fh_assemble_source_vector(fh_ctx*, uint32_t solution_dim, const double* g, const double* values, double* out);
int fh_assemble_source_vector_dev(fh_ctx*, uint32_t solution_dim, const double* g /* host */, const double* values_dev,
                                  double* out_dev);
int fh_physical_quadrature_points(fh_ctx*, double* x /* E x nq x d */);
int fh_physical_quadrature_points_dev(fh_ctx*, double* x_dev);
/* single element matrix, (s n)^2 column-major: ElementMatrixAssembler::assemble_element_matrix_into
 * (src/assembly/local.rs:78, elliptic.rs:299-340) -- for unit tests of the element kernels */
int fh_assemble_element_matrices(fh_ctx*, uint64_t first_element, uint64_t count, double* ke_out);
int fh_assemble_element_matrices_dev(fh_ctx*, uint64_t first_element, uint64_t count, double* ke_out_dev);

/* ---- post-assembly helpers (callers of the path) ------------------------------------------------ */
/* apply_homogeneous_dirichlet_bc_csr / _rhs (global.rs:379-451, 479-495) on device-resident CSR */
int fh_apply_dirichlet_csr_dev(fh_ctx*, double* values_dev, const uint64_t* nodes, uint64_t num_nodes);
int fh_apply_dirichlet_rhs_dev(fh_ctx*, double* rhs_dev, const uint64_t* nodes, uint64_t num_nodes);

/* ---- host-side input generators (no device needed) -------------------------------------------- */
/* fenris-quadrature/src/univariate.rs:66-118, tensor.rs:13-55 */
int fh_gauss(uint32_t n, double* weights, double* points);
int fh_quadrilateral_gauss(uint32_t n, double* weights, double* points);
int fh_hexahedron_gauss(uint32_t n, double* weights, double* points);
/* polyquad tables (fenris-quadrature/rules/polyquad/expanded/{tet,tri}); returns FH_UNSUPPORTED for
 * strengths that are not tabulated here; *num_points receives the rule size */
int fh_tetrahedron_rule(uint32_t strength, double* weights, double* points, uint32_t* num_points);
int fh_triangle_rule(uint32_t strength, double* weights, double* points, uint32_t* num_points);
/* src/mesh/procedural.rs:46-93, 216-277, 286-403.  Sizes first (vertices/cells may be NULL). */
int fh_quad_mesh_2d(double unit_length, uint64_t units_x, uint64_t units_y, uint64_t cells_per_unit,
                    const double top_left[2], double* vertices, uint64_t* connectivity,
                    uint64_t* num_vertices, uint64_t* num_cells);
int fh_hex_mesh(double unit_length, uint64_t units_x, uint64_t units_y, uint64_t units_z, uint64_t cells_per_unit,
                double* vertices, uint64_t* connectivity, uint64_t* num_vertices, uint64_t* num_cells);
int fh_tet_mesh(double unit_length, uint64_t units_x, uint64_t units_y, uint64_t units_z, uint64_t cells_per_unit,
                double* vertices, uint64_t* connectivity, uint64_t* num_vertices, uint64_t* num_cells);
/* Hex27Mesh::from(&hex8_mesh) (src/mesh_convert.rs:85-166, 227-330).  out_vertices capacity
 * 27*num_cells*3 doubles, out_connectivity 27*num_cells. */
int fh_hex8_to_hex27(const double* vertices, uint64_t num_vertices, const uint64_t* hex8, uint64_t num_cells,
                     double* out_vertices, uint64_t* out_num_vertices, uint64_t* out_connectivity);
/* p-refinement of the linear meshes (src/mesh_convert.rs): Tet10Mesh::from(&tet4) (:42-83, 444-452: vertex nodes then
 * the edge nodes (0,1) (1,2) (0,2) (0,3) (2,3) (1,3), labels in order of first occurrence), Tri6 from Tri3 (:332-383)
 * and Quad9 from Quad4 (:385-442): the old vertices keep their indices, edge midpoints are appended in order of first
 * occurrence, Quad9 also appends the cell midpoint.  from_kind is FH_TET4 / FH_TRI3 / FH_QUAD4, FH_HEX8 for
 * Hex20Mesh::from(&hex8) (:168-217: the 8 vertex nodes and the 12 edge nodes of Hex27, same labelling); out_vertices capacity
 * (nodes per refined element) * num_cells * d doubles. */
/* Tet20Mesh::from(&tet4) (src/mesh_convert.rs:658-775): new vertices = the sorted, deduplicated 4-tuples [idx,0,0,0]
 * (vertex), [min,max,local,1] (two nodes per edge, local counted from min), [a,b,c,2] (face centroid); the label of a
 * vertex is its rank in that order.  out_vertices capacity 20 * num_cells * 3 doubles. */
int fh_tet4_to_tet20(const double* vertices, uint64_t num_vertices, const uint64_t* tet4, uint64_t num_cells,
                     double* out_vertices, uint64_t* out_num_vertices, uint64_t* out_connectivity);
int fh_refine_to_quadratic(int from_kind, const double* vertices, uint64_t num_vertices, const uint64_t* connectivity,
                           uint64_t num_cells, double* out_vertices, uint64_t* out_num_vertices, uint64_t* out_connectivity);
/* load_msh_from_bytes (src/io/msh.rs:47-111), Gmsh MSH 4.1 ASCII: vertices of all node blocks in file order (x, y for the
 * 2-D kinds), the elements of every block whose (Gmsh element type, entity dimension) matches elem_kind, node tags
 * minus one, no reordering.  Two-phase: call with vertices = connectivity = NULL for the sizes.  Blocks with
 * non-consecutive tags are refused like in the reference; binary files are not supported (FH_BAD_ARGUMENT, message
 * from fh_msh_last_error, thread-local). */
int fh_load_msh(const char* bytes, uint64_t len, int elem_kind, double* vertices, uint64_t* num_vertices,
                uint64_t* connectivity, uint64_t* num_elements);
const char* fh_msh_last_error(void);
/* cuthill_mckee on a square sparsity pattern (src/mesh/reorder.rs:171-233): perm_out[target] = source.  The
 * reference orders equal-degree neighbours with an unstable sort (unspecified); ties are broken by ascending index. */
int fh_cuthill_mckee(uint64_t num_rows, const uint64_t* row_offsets, const uint64_t* col_indices, uint64_t* perm_out);
/* reorder_mesh_par (src/mesh/reorder.rs:54-95): reverse Cuthill-McKee vertex permutation + elements sorted by their
 * smallest new vertex index.  vertex_perm[new] = old, connectivity_perm[new] = old; MeshPermutation::apply
 * (reorder.rs:29-52) relabels the vertex indices inside the elements with the inverse vertex permutation. */
int fh_reorder_mesh(uint64_t num_vertices, uint64_t nodes_per_element, const uint64_t* connectivity, uint64_t num_elements,
                    uint64_t* vertex_perm, uint64_t* connectivity_perm);
/* LameParameters::from(YoungPoisson) (fenris-solid/src/materials.rs:31-43) */
int fh_lame_from_young_poisson(double young, double poisson, double* mu, double* lambda);

/* ---- introspection for benchmarks --------------------------------------------------------------- */
/* name of the device kernel the last fh_assemble_matrix* call launched (for rocprof matching) */
const char* fh_last_kernel_name(const fh_ctx*);

/* ---- solve + error estimation on the device-resident system (what every caller does next) ------------ */
/* y = K x on the CSR of the context's pattern: LinearOperator for CsrMatrix (fenris-sparse/src/cg.rs:44-52) */
int fh_spmv_dev(fh_ctx*, const double* values_dev, const double* x_dev, double* y_dev);
/* ConjugateGradient::solve_with_guess (fenris-sparse/src/cg.rs:366-478) with RelativeResidualCriterion(rel_tol)
 * (:86-124, the solver's own residual, ||r|| <= tol ||b||) and, for FH_PRECOND_JACOBI, the inverse diagonal as
 * the preconditioner -- exactly how solve_linear_system drives it (tests/convergence_tests/
 * poisson_mms_common.rs:142-163: max_iter 10000, tol 1e-9).  x holds the initial guess on entry and the solution on
 * return (the iterate reached so far on failure); max_iter == 0 means no limit; *num_iterations counts the updates
 * of x.  Errors: FH_CG_MAX_ITERATIONS, FH_CG_INDEFINITE_OPERATOR (p.Ap <= 0), FH_CG_INDEFINITE_PRECONDITIONER
 * (z.r <= 0).  Reductions are ordered (no floating-point atomics): runs are bitwise reproducible. */
enum { FH_PRECOND_IDENTITY = 0, FH_PRECOND_JACOBI = 1 };
int fh_cg_solve(fh_ctx*, const double* values, const double* b, double* x, int preconditioner, double rel_tol,
                uint64_t max_iter, uint64_t* num_iterations);
int fh_cg_solve_dev(fh_ctx*, const double* values_dev, const double* b_dev, double* x_dev, int preconditioner, double rel_tol,
                    uint64_t max_iter, uint64_t* num_iterations);
/* estimate_L2_error_squared / estimate_H1_seminorm_error_squared (src/error.rs:287-372):
 *   sum_e sum_q w |det J| |u_h(x_q) - u(x_q)|^2      resp.   |grad u_h(x_q) - grad u(x_q)|_F^2
 * with the quadrature table of the context.  The reference solution is arbitrary code in the reference; here the
 * caller samples it at the physical points of fh_physical_quadrature_points: u_exact is (E, nq, s); grad_exact is
 * (E, nq, d, s) with grad[i][k] = d u_k / d x_i.  The H1 form inverts J: FH_SINGULAR_JACOBIAN if det J == 0. */
int fh_estimate_L2_error_squared(fh_ctx*, uint32_t solution_dim, const double* u_h, const double* u_exact, double* out);
int fh_estimate_L2_error_squared_dev(fh_ctx*, uint32_t solution_dim, const double* u_h_dev, const double* u_exact_dev, double* out);
int fh_estimate_H1_seminorm_error_squared(fh_ctx*, uint32_t solution_dim, const double* u_h, const double* grad_exact, double* out);
int fh_estimate_H1_seminorm_error_squared_dev(fh_ctx*, uint32_t solution_dim, const double* u_h_dev, const double* grad_exact_dev,
                                              double* out);

/* ---- composition of element assemblers: AggregateElementAssembler, MapElementNodes and TransformElementMatrix / Vector with
 * a scale factor (src/assembly/local.rs:152-340; tests/unit_tests/assembly/local.rs:189-336).  Every body keeps its own
 * context -- mesh, operator, table, element kind, fastest kernels -- and what it assembled in ITS node numbering is added,
 * scaled, into a matrix / vector over the aggregate's node index space:
 *     dst(map[i] s + r, map[j] s + c) += scale * src(i s + r, j s + c)        dst(map[i] s + r) += scale * src(i s + r)
 * node_map_dev: num_nodes(ctx) u64 on the device, or NULL for the identity (an aggregate over one shared node space).
 * The destination pattern (scalar CSR, u64, on the device; e.g. fh_pattern of a context that holds the aggregate's ragged
 * connectivity, fh_set_connectivity_ragged) must hold every mapped entry: a node out of range or a missing column is
 * FH_BAD_ARGUMENT (the reference panics, global.rs:531-533).  src/dst values on the device, same solution dimension. */
int fh_add_mapped_matrix_dev(fh_ctx* ctx, const double* src_values_dev, const uint64_t* node_map_dev, double scale,
                             uint64_t dst_num_nodes, const uint64_t* dst_row_offsets_dev, const uint64_t* dst_col_indices_dev,
                             double* dst_values_dev);
int fh_add_mapped_vector_dev(fh_ctx* ctx, const double* src_dev, const uint64_t* node_map_dev, double scale,
                             uint64_t dst_num_nodes, double* dst_dev);
/* The same with the solution dimension given by the caller (> 0): for the vector of an ElementSourceAssembler body, whose
 * context holds a mesh and a quadrature table but no operator (local/source.rs:159-278; the dimension is the source's). */
int fh_add_mapped_vector_sdim_dev(fh_ctx* ctx, const double* src_dev, const uint64_t* node_map_dev, double scale, int solution_dim,
                                  uint64_t dst_num_nodes, double* dst_dev);

/* ---- multi-GPU: one process or thread per GPU, each with its own fh_ctx holding one partition (its own
 * elements plus the halo layers whose nodes it shares, so that interface rows have the global pattern and the same layout
 * on both sides; numerics over the own elements: fh_set_active_elements).  The only data that crosses a partition boundary
 * are the partial rows of interface nodes: the non-owner sends its contiguous segment of `values` to the owner, which adds
 * it -- RCCL point-to-point (ncclSend / ncclRecv) on a side stream, each interface on its own xGMI link, beside the
 * assembly launches; never a collective over the matrix.  Replaces, across partitions, what the shared address space does
 * for CsrParAssembler::assemble_into_csr (src/assembly/global.rs:314-376).
 * Bootstrap like NCCL: rank 0 draws an id, the host distributes its bytes to every rank by whatever means it has. */
typedef struct fh_group fh_group;
#define FH_GROUP_ID_BYTES 128
int fh_group_unique_id(uint8_t id[FH_GROUP_ID_BYTES]);
/* collective over all `world` ranks; `ctx` gives the device and the stream the exchange is ordered against */
int fh_group_create(fh_ctx* ctx, const uint8_t id[FH_GROUP_ID_BYTES], int rank, int world, fh_group** out);
/* A group refers to its context: destroy every group of a context BEFORE fh_destroy(ctx). */
void fh_group_destroy(fh_group*);
/* number of ranks of the communicator as RCCL reports it (ncclCommCount) */
int fh_group_size(const fh_group*, int* ranks);
/* What this rank moves in one exchange: values[send_first .. send_first + send_count) go to rank send_peer (-1: nothing);
 * recv_count values arrive from rank recv_peer (-1: nothing) and are ADDED to values[recv_first ..).  For z-slabs:
 * send = rows of the bottom ghost plane to rank - 1, receive = rows of the owned top plane from rank + 1. */
int fh_group_set_exchange(fh_group*, int send_peer, uint64_t send_first, uint64_t send_count, int recv_peer,
                          uint64_t recv_first, uint64_t recv_count);
/* start: call after enqueueing the launch that produces the rows to send (fh_assemble_matrix_rows_async_dev for the
 * interface rows); the transfers run on the group's stream while the caller enqueues the rest of the assembly.
 * finish: the context's stream waits for the transfers and adds the received rows.  Errors: fh_last_error(ctx). */
int fh_group_exchange_start(fh_group*, double* values_dev);
int fh_group_exchange_finish(fh_group*, double* values_dev);
/* Arbitrary partitions (an element partition elem_to_part[] of an unstructured mesh: no planes, any number of neighbours, interface
 * rows anywhere): per peer a list of LOCAL NODES whose rows (all S scalar rows of the node, as they lie in `values`) are packed and sent,
 * and a list of owned local nodes whose rows receive-and-add what the peer packed -- both sides list the shared nodes in the same
 * (ascending global) order, so no indices travel.  Offsets have peers + 1 entries.  The context's pattern must have been built.  All
 * transfers of one exchange are posted in one RCCL group; fh_group_exchange_start / _finish drive this mode once it is set.  A rank
 * may list itself as a peer (a device copy inside RCCL: the single-GPU test does).  The nodes of ONE peer's list must be distinct; a node
 * may appear in the lists of several peers (their rows are added peer by peer, in list order).
 * CONTRACT: the exchange ships the listed rows AS THEY STAND and the receiver ADDS them.  The rows a rank sends must therefore hold this
 * assembly's partial sums only -- assemble them with FH_ASSEMBLE_OVERWRITE (the reference's accumulate-into-the-output semantics,
 * global.rs:133-182, applies to the OWNED rows after the exchange, not to the rows in transit): earlier content of a sent row would be
 * counted once more on its owner for every rank that sends it, without any error. */
int fh_group_set_exchange_nodes(fh_group*, int num_send_peers, const int32_t* send_peers, const uint64_t* send_offsets,
                                const uint64_t* send_nodes, int num_recv_peers, const int32_t* recv_peers, const uint64_t* recv_offsets,
                                const uint64_t* recv_nodes);

/* Node VECTORS through the same lists (the residual / source vector of a partition: fh_assemble_vector_dev over the active elements
 * leaves partial sums at the nodes other ranks own): `components` values per node (the solution dim), packed, sent, received and added
 * exactly like the rows above.  Entries of nodes the rank does not own are scratch afterwards.
 * CONTRACT (as for the rows): `vec_dev` must hold THIS assembly's partial sums only -- assemble into a zeroed vector, exchange, then add the
 * owned entries to whatever they accumulate into (fenris_amd/partition.py: PartAssembly.assemble_vector does exactly that). */
int fh_group_exchange_vector_start(fh_group*, double* vec_dev, uint32_t components);
int fh_group_exchange_vector_finish(fh_group*, double* vec_dev, uint32_t components);

/* ---- element partitions of arbitrary meshes (host; SURVEY.md 8e "the engine takes elem_to_part[]").  What one rank needs to run
 * its share of CsrParAssembler::assemble_into_csr (global.rs:314-376): a node is owned by the LOWEST part that touches it; the
 * extended local mesh holds the rank's own elements plus every element touching a node they touch, numbered by ascending global id
 * (the rows of every node the rank contributes to then carry the GLOBAL pattern in the global column order); numerics over the own
 * elements (fh_set_active_elements); interface rows through fh_group_set_exchange_nodes.  fenris_amd/partition.py mirrors this. */
typedef struct fh_partition fh_partition;
/* default partitioner: Morton order of the element centroids cut into `world` equal runs */
int fh_morton_partition(uint32_t dim, const double* vertices, uint64_t num_vertices, uint64_t nodes_per_element,
                        const uint64_t* connectivity, uint64_t num_elements, uint32_t world, int32_t* elem_to_part);
/* NULL on a bad argument (a part outside [0, world), a node index >= num_nodes).  halo_mode != 0: the halo elements that touch an owned
 * node are active as well -- the owned rows are complete without any exchange (the lists stay empty). */
fh_partition* fh_partition_create(uint64_t num_nodes, uint64_t nodes_per_element, const uint64_t* connectivity, uint64_t num_elements,
                                  const int32_t* elem_to_part, int rank, int world, int halo_mode);
void fh_partition_destroy(fh_partition*);
/* sizes: local nodes, local elements, owned nodes, own elements, peers sent to, nodes sent, peers received from, nodes received */
int fh_partition_sizes(const fh_partition*, uint64_t sizes[8]);
/* global id of every local node / element, connectivity in local node ids, element mask, owned local nodes (NULL skips an array) */
int fh_partition_mesh(const fh_partition*, uint64_t* l2g, uint64_t* elem_l2g, uint64_t* local_connectivity, uint8_t* active,
                      uint64_t* owned_nodes);
/* the arguments of fh_group_set_exchange_nodes (NULL skips an array) */
int fh_partition_exchange(const fh_partition*, int32_t* send_peers, uint64_t* send_offsets, uint64_t* send_nodes, int32_t* recv_peers,
                          uint64_t* recv_offsets, uint64_t* recv_nodes);

#ifdef __cplusplus
}
#endif
#endif /* FENRIS_HIP_H */
