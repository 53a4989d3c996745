// Shared device-side definitions of the assembly engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fenris_hip.h"

namespace fenris_hip {

// ------------------------------------------------------------------------------------ element traits
// D = geometry/reference dim, N = nodes, NG = nodes of the geometry map (Hex27 is sub-parametric: its
// Jacobian is the trilinear map of its first 8 nodes, src/element/hexahedron.rs:324-326).
template <int EK> struct ElemT;
template <> struct ElemT<FH_QUAD4> { static constexpr int D = 2, N = 4, NG = 4; };
template <> struct ElemT<FH_HEX8>  { static constexpr int D = 3, N = 8, NG = 8; };
template <> struct ElemT<FH_TET4>  { static constexpr int D = 3, N = 4, NG = 4; };
template <> struct ElemT<FH_HEX27> { static constexpr int D = 3, N = 27, NG = 8; };
template <> struct ElemT<FH_TRI3>  { static constexpr int D = 2, N = 3, NG = 3; };
template <> struct ElemT<FH_TET10> { static constexpr int D = 3, N = 10, NG = 4; };
template <> struct ElemT<FH_QUAD9> { static constexpr int D = 2, N = 9, NG = 4; };
template <> struct ElemT<FH_TRI6>  { static constexpr int D = 2, N = 6, NG = 3; };
template <> struct ElemT<FH_HEX20> { static constexpr int D = 3, N = 20, NG = 8; };
template <> struct ElemT<FH_TET20> { static constexpr int D = 3, N = 20, NG = 4; };

// ------------------------------------------------------------------------------------ operator traits
// Per (element, quadrature point) the prologue leaves in LDS:  NVEC vectors per node (physical
// gradient g_n; for NeoHookean also F^{-T} g_n; for StVK also F g_n and E g_n) and NCOEF scalars.
template <int OP, int D> struct OpT;
template <int D> struct OpT<FH_LAPLACE, D> {
    static constexpr int S = 1, NVEC = 1, NCOEF = 1;  // [s]
    static constexpr bool NEEDS_U = false;
};
template <int D> struct OpT<FH_LINEAR_ELASTIC, D> {
    static constexpr int S = D, NVEC = 1, NCOEF = 2;  // [s*mu, s*lambda]
    static constexpr bool NEEDS_U = false;
};
template <int D> struct OpT<FH_NEO_HOOKEAN, D> {
    static constexpr int S = D, NVEC = 2, NCOEF = 3;  // [s*lambda, s*alpha, s*mu]
    static constexpr bool NEEDS_U = true;
};
template <int D> struct OpT<FH_STVK, D> {
    static constexpr int S = D, NVEC = 3, NCOEF = 4 + D * D;  // [2 s mu, s lambda trE, s mu, s lambda, F F^T]
    static constexpr bool NEEDS_U = true;
};

// mass matrix (src/assembly/local/mass.rs): the node "vector" slot carries phi_n in its first component
template <int D> struct OpT<FH_MASS_SCALAR, D> {
    static constexpr int S = 1, NVEC = 1, NCOEF = 1;  // [s * rho]
    static constexpr bool NEEDS_U = false;
};
template <int D> struct OpT<FH_MASS_VECTOR, D> {
    static constexpr int S = D, NVEC = 1, NCOEF = 1;
    static constexpr bool NEEDS_U = false;
};

// operator given as data (fenris_hip.h, FH_TENSOR): one d x d x d x d tensor per quadrature point from KArgs::tensor; the point record is the
// physical gradients and the scale, like Laplace's
template <int D> struct OpT<FH_TENSOR, D> {
    static constexpr int S = D, NVEC = 1, NCOEF = 1;  // [s]
    static constexpr bool NEEDS_U = false;
};

enum { MODE_ATOMIC = 0, MODE_COLORED = 1, MODE_GATHER = 2, MODE_DUMP = 3 };

// status words in device memory
struct DevStatus {
    int singular;                    // set to 1 if any Jacobian determinant was exactly zero
    int pad;
    unsigned long long failed_elem;  // lowest failing element (atomicMin)
};

// per-block header of the owner-computes (gather) tables
struct GatherHdr {
    int i0, nb;      // first node, number of nodes
    int r0, nrow;    // first node-level CSR entry, number of node-level entries of the block's rows
    int k0, m;       // first (node, element) entry in n2e, number of entries
    int u_off, U;    // offset into gt_elems, number of unique adjacent elements
};

// kernel arguments (plain struct, passed by value)
struct KArgs {
    // mesh
    const double* verts;   // N x D
    const int* conn;       // E x n
    long long num_elements;
    int num_nodes;
    // quadrature tables (device)
    int nq;
    int ablate;            // profiling only (FENRIS_HIP_ABLATE bitmask): 1 skip phase B, 2 skip phase C, 4 skip finalize, 8 skip phase D,
                           // 16 write-out behind the barrier instead of overlapped, 32 plain LDS stores instead of atomics (wrong sums)
    int fast;              // uniform parameters and non-negative weights: sqrt-scaled gradients
    double mu, lambda;     // uniform Lame parameters (fast path)
    const double* qw;      // nq
    const double* gref;    // nq x N x D   reference gradients of the element basis
    const double* gref_t;  // N x nq x D   the same table node-major (Hex27: hex27_blocks.hpp), or null
    unsigned long long vtx_pack;  // hex27_blocks.hpp: conn, gref and gref_t have their node index permuted; 5 bits per geometry vertex = its place
    const double* ggeom;   // nq x NG x D  reference gradients of the geometry map
    const double* phiref;  // nq x N       basis values (mass matrix only)
    int all_affine;        // Hex8: every element is a parallelepiped by k_classify_affine_hex8's test (the element pass takes J as constant without testing)
    const double* qmono;   // nq x 8       Hex8 only, or null: xi eta zeta, eta zeta, xi zeta, xi eta of every point (element_pass.hpp, monomial form)
    const double* qmom;    // 8            Hex8 only, or null: moments of the rule (sum w, xi^2, eta^2, zeta^2, eta^2 zeta^2, xi^2 zeta^2, xi^2 eta^2) when every
                           //              moment with an odd power vanishes AND the parameters are the same at every point (element_pass.hpp, AFFM = 2)
    const double* qparams; // nq x 2 (mu, lambda) or null
    const double* tensor;  // FH_TENSOR: nq x d^4 coefficient tensors, index ((i d + j) d + k) d + l
    int nonsym;            // FH_TENSOR: bit 0 = Symmetry::NonSymmetric (every block formed, nothing mirrored); bit 1 (MODE_DUMP, two-pass assembly):
                           // store K_e TRANSPOSED, so that the row gather's contiguous "columns" are the rows of K_e
    // CompactQuadratureTable with shared points / weights: element e reads rparams[(rule_map[e] nq + q) 2 ..]
    const unsigned* rule_map;  // E, or null (uniform table)
    const double* rparams;     // num_rules x nq x 2
    const double* u;       // S x N or null
    // node-level pattern
    const unsigned* noff;     // N+1
    const unsigned* ncols;    // nnz_n
    const unsigned* n2e_off;  // N+1
    const unsigned* n2e;      // E*n entries e*n+a, ascending per node
    // output
    double* vals;
    double* vec_out;
    double* scalar_out;
    double* ke_out;
    int ke_by_elem;          // MODE_DUMP: index ke_out by element id instead of by work position
    int ke_tri;              // MODE_DUMP (two-pass assembly, s = 3): the LOWER node-block triangle, row-major, block (J, I <= J) = 9 contiguous doubles
    int overwrite;
    DevStatus* status;
    unsigned long long* trace;  // profiling only: 7 counters (6 phase cycle sums over waves, wave count) or null
    // element-centric work description
    const unsigned* labels;  // element list (colour) or null for identity
    long long work_begin, work_end;
    int epb;                 // elements per block
    int nc_row;              // > 0: stage the neighbour lists (longest has nc_row entries) of the unit's nodes in LDS
    // gather work description
    const unsigned* blk_off; // node block boundaries, nblk+1
    const GatherHdr* gt_hdr; // per block
    const unsigned* gt_elems;// unique adjacent elements of each block
    const unsigned* gt_ent;  // per n2e entry: unique slot << 16 | local index << 8 | block-local node
    const unsigned char* gt_pos; // per (n2e entry, local node): column slot in the owner's row (or null)
    int nblk;
    int ub;                  // max unique elements staged at a time
    int mb;                  // max (node, element) entries per batch (gather)
    int acc_max;             // doubles reserved for row accumulators
    int nb_max;              // max nodes per block
};

__device__ __forceinline__ void report_singular(DevStatus* st, long long e) {
    st->singular = 1;
    atomicMin(&st->failed_elem, static_cast<unsigned long long>(e));
}

// hardware fp64 atomic add (global_atomic_add_f64 / ds_add_f64); never a CAS loop
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

}  // namespace fenris_hip
