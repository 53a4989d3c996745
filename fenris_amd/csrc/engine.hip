// fenris_hip engine: context, device memory, pattern build, dispatch of the assembly kernels and
// the C ABI declared in include/fenris_hip.h.  gfx950 only.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/fenris_hip.h"
#include "assemble_kernels.hpp"
#include "solver_kernels.hpp"
#include "hex27_mfma.hpp"
#include "rows_kernel.hpp"
#include "affine_kernel.hpp"
#include "affine_rows.hpp"
#include "element_pass.hpp"
#include "coloring_kernels.hpp"
#include "hex8_rows.hpp"
#include "vector_tiles.hpp"
#include "device_common.hpp"
#include "group_internal.hpp"
#include "host_inputs.hpp"
#include "pattern_kernels.hpp"

extern char** environ;

using namespace fenris_hip;

namespace {

// ------------------------------------------------------------------------------------------------
// small RAII device buffer
// ------------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
};

// reference gradient tables (host): product-side evaluation of the shape-function gradients.
// Node sign tables and 1-D factors: SURVEY Appendix A.1/A.2 (src/element.rs:244-298,
// hexahedron.rs:49-58, 229-264, quadrilateral.rs:84-99, tetrahedron.rs:561-568, triangle.rs:82-89).
const double HEX_SIGN[27][3] = {
    {-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1},
    {0, -1, -1}, {-1, 0, -1}, {-1, -1, 0}, {1, 0, -1}, {1, -1, 0}, {0, 1, -1}, {1, 1, 0}, {-1, 1, 0},
    {0, -1, 1}, {-1, 0, 1}, {1, 0, 1}, {0, 1, 1},
    {0, 0, -1}, {0, -1, 0}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
const double QUAD_SIGN[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
const double QUAD9_SIGN[9][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}, {0, -1}, {1, 0}, {0, 1}, {-1, 0}, {0, 0}};
void ref_basis(int kind, const double* xi, double* out);

inline double lin(double al, double x) { return (1.0 + al * x) / 2.0; }
inline double dlin(double al) { return al / 2.0; }
inline double quad(double al, double x) { const double a2 = al * al; return (3.0 / 2.0 * a2 - 1.0) * (x * x) + 0.5 * al * x + 1.0 - a2; }
inline double dquad(double al, double x) { const double a2 = al * al; return 2.0 * (3.0 / 2.0 * a2 - 1.0) * x + 0.5 * al; }

// out: n x d (node-major, AoS per node)
void ref_gradients(int kind, const double* xi, double* out) {
    switch (kind) {
        case FH_QUAD4:
            for (int n = 0; n < 4; ++n) {
                const double al = QUAD_SIGN[n][0], be = QUAD_SIGN[n][1];
                out[2 * n] = al * (1.0 + be * xi[1]) / 4.0;
                out[2 * n + 1] = be * (1.0 + al * xi[0]) / 4.0;
            }
            break;
        case FH_HEX8:
            for (int n = 0; n < 8; ++n) {
                const double* s = HEX_SIGN[n];
                out[3 * n] = dlin(s[0]) * lin(s[1], xi[1]) * lin(s[2], xi[2]);
                out[3 * n + 1] = lin(s[0], xi[0]) * dlin(s[1]) * lin(s[2], xi[2]);
                out[3 * n + 2] = lin(s[0], xi[0]) * lin(s[1], xi[1]) * dlin(s[2]);
            }
            break;
        case FH_HEX27:
            for (int n = 0; n < 27; ++n) {
                const double* s = HEX_SIGN[n];
                out[3 * n] = dquad(s[0], xi[0]) * quad(s[1], xi[1]) * quad(s[2], xi[2]);
                out[3 * n + 1] = quad(s[0], xi[0]) * dquad(s[1], xi[1]) * quad(s[2], xi[2]);
                out[3 * n + 2] = quad(s[0], xi[0]) * quad(s[1], xi[1]) * dquad(s[2], xi[2]);
            }
            break;
        case FH_TET4: {
            static const double G[12] = {-0.5, -0.5, -0.5, 0.5, 0, 0, 0, 0.5, 0, 0, 0, 0.5};
            std::memcpy(out, G, sizeof G);
            break;
        }
        case FH_TRI3: {
            static const double G[6] = {-0.5, -0.5, 0.5, 0, 0, 0.5};
            std::memcpy(out, G, sizeof G);
            break;
        }
        case FH_TET10:
        case FH_TRI6: {
            // vertex node i: g_i (4 psi_i - 1); edge node (i, j): 4 (g_i psi_j + g_j psi_i), psi / g of the linear
            // simplex (tetrahedron.rs:198-224, triangle.rs:228-252)
            const int lin_kind = (kind == FH_TET10) ? FH_TET4 : FH_TRI3;
            const int d = (kind == FH_TET10) ? 3 : 2, nv = d + 1;
            static const int E3[6][2] = {{0, 1}, {1, 2}, {0, 2}, {0, 3}, {2, 3}, {1, 3}};
            static const int E2[3][2] = {{0, 1}, {1, 2}, {0, 2}};
            double psi[4], g[12];
            ref_basis(lin_kind, xi, psi);
            ref_gradients(lin_kind, xi, g);
            for (int i = 0; i < nv; ++i)
                for (int k = 0; k < d; ++k) out[d * i + k] = g[d * i + k] * (4.0 * psi[i] - 1.0);
            const int ne = (kind == FH_TET10) ? 6 : 3;
            for (int m = 0; m < ne; ++m) {
                const int i = (kind == FH_TET10) ? E3[m][0] : E2[m][0], j = (kind == FH_TET10) ? E3[m][1] : E2[m][1];
                for (int k = 0; k < d; ++k) out[d * (nv + m) + k] = g[d * i + k] * (4.0 * psi[j]) + g[d * j + k] * (4.0 * psi[i]);
            }
            break;
        }
        case FH_TET20: {  // tetrahedron.rs:404-466: products of the Tet4 basis psi and its gradients g
            static const int ED[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
            static const int FA[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
            double psi[4], g[12];
            ref_basis(FH_TET4, xi, psi);
            ref_gradients(FH_TET4, xi, g);
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 3; ++k) out[3 * i + k] = g[3 * i + k] * 0.5 * (27.0 * psi[i] * psi[i] - 18.0 * psi[i] + 2.0);
            for (int m = 0; m < 6; ++m)
                for (int half = 0; half < 2; ++half) {  // edge_gradient(a, b): the node closer to a
                    const int a = half ? ED[m][1] : ED[m][0], b = half ? ED[m][0] : ED[m][1];
                    const double pa = psi[a], pb = psi[b];
                    for (int k = 0; k < 3; ++k)
                        out[3 * (4 + 2 * m + half) + k] = (g[3 * a + k] * (pb * (6.0 * pa - 1.0)) + g[3 * b + k] * (pa * (3.0 * pa - 1.0))) * (9.0 / 2.0);
                }
            for (int f = 0; f < 4; ++f) {
                const int a = FA[f][0], b = FA[f][1], c = FA[f][2];
                for (int k = 0; k < 3; ++k)
                    out[3 * (16 + f) + k] = (g[3 * a + k] * psi[b] * psi[c] + g[3 * b + k] * psi[a] * psi[c] + g[3 * c + k] * psi[a] * psi[b]) * 27.0;
            }
            break;
        }
        case FH_HEX20:
            for (int n = 0; n < 20; ++n) {  // hexahedron.rs:465-543: phi = s f g (corners) / s h g (edges), product rule
                const double al = HEX_SIGN[n][0], be = HEX_SIGN[n][1], ga = HEX_SIGN[n][2];
                const double ax = 1.0 + al * xi[0], by = 1.0 + be * xi[1], cz = 1.0 + ga * xi[2];
                const double g = ax * by * cz;
                if (n < 8) {
                    const double f = al * xi[0] + be * xi[1] + ga * xi[2] - 2.0, s = 1.0 / 8.0;
                    out[3 * n] = s * (al * g + f * al * by * cz);
                    out[3 * n + 1] = s * (be * g + f * be * ax * cz);
                    out[3 * n + 2] = s * (ga * g + f * ga * ax * by);
                } else {
                    const double a2 = al * al, b2 = be * be, c2 = ga * ga, s = 1.0 / 4.0;
                    const double hx = 1.0 - (1.0 - a2) * xi[0] * xi[0], hy = 1.0 - (1.0 - b2) * xi[1] * xi[1], hz = 1.0 - (1.0 - c2) * xi[2] * xi[2];
                    const double h = hx * hy * hz;
                    const double dh0 = -2.0 * (1.0 - a2) * xi[0] * hy * hz, dh1 = -2.0 * (1.0 - b2) * xi[1] * hx * hz,
                                 dh2 = -2.0 * (1.0 - c2) * xi[2] * hx * hy;
                    out[3 * n] = s * (dh0 * g + h * al * by * cz);
                    out[3 * n + 1] = s * (dh1 * g + h * be * ax * cz);
                    out[3 * n + 2] = s * (dh2 * g + h * ga * ax * by);
                }
            }
            break;
        case FH_QUAD9:
            for (int n = 0; n < 9; ++n) {  // quadrilateral.rs:280-313
                const double al = QUAD9_SIGN[n][0], be = QUAD9_SIGN[n][1];
                out[2 * n] = quad(be, xi[1]) * dquad(al, xi[0]);
                out[2 * n + 1] = quad(al, xi[0]) * dquad(be, xi[1]);
            }
            break;
    }
}

// out: n basis values (src/element: quadrilateral.rs:79-90, hexahedron.rs:43-59, 222-265, tetrahedron.rs:551-558,
// triangle.rs:72-78)
void ref_basis(int kind, const double* xi, double* out) {
    switch (kind) {
        case FH_QUAD4:
            for (int n = 0; n < 4; ++n) out[n] = (1.0 + QUAD_SIGN[n][0] * xi[0]) * (1.0 + QUAD_SIGN[n][1] * xi[1]) / 4.0;
            break;
        case FH_HEX8:
            for (int n = 0; n < 8; ++n) out[n] = lin(HEX_SIGN[n][0], xi[0]) * lin(HEX_SIGN[n][1], xi[1]) * lin(HEX_SIGN[n][2], xi[2]);
            break;
        case FH_HEX27:
            for (int n = 0; n < 27; ++n) out[n] = quad(HEX_SIGN[n][0], xi[0]) * quad(HEX_SIGN[n][1], xi[1]) * quad(HEX_SIGN[n][2], xi[2]);
            break;
        case FH_TET4:
            out[0] = -0.5 * xi[0] - 0.5 * xi[1] - 0.5 * xi[2] - 0.5;
            out[1] = 0.5 * xi[0] + 0.5;
            out[2] = 0.5 * xi[1] + 0.5;
            out[3] = 0.5 * xi[2] + 0.5;
            break;
        case FH_TRI3:
            out[0] = -0.5 * xi[0] - 0.5 * xi[1];
            out[1] = 0.5 * xi[0] + 0.5;
            out[2] = 0.5 * xi[1] + 0.5;
            break;
        case FH_TET10: {  // tetrahedron.rs:179-195
            double p[4];
            ref_basis(FH_TET4, xi, p);
            for (int i = 0; i < 4; ++i) out[i] = p[i] * (2.0 * p[i] - 1.0);
            out[4] = 4.0 * p[0] * p[1]; out[5] = 4.0 * p[1] * p[2]; out[6] = 4.0 * p[0] * p[2];
            out[7] = 4.0 * p[0] * p[3]; out[8] = 4.0 * p[2] * p[3]; out[9] = 4.0 * p[1] * p[3];
            break;
        }
        case FH_TRI6: {  // triangle.rs:211-224
            double p[3];
            ref_basis(FH_TRI3, xi, p);
            for (int i = 0; i < 3; ++i) out[i] = p[i] * (2.0 * p[i] - 1.0);
            out[3] = 4.0 * p[0] * p[1]; out[4] = 4.0 * p[1] * p[2]; out[5] = 4.0 * p[0] * p[2];
            break;
        }
        case FH_TET20: {  // tetrahedron.rs:346-401
            static const int ED[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
            static const int FA[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
            double psi[4];
            ref_basis(FH_TET4, xi, psi);
            for (int i = 0; i < 4; ++i) out[i] = 0.5 * psi[i] * (3.0 * psi[i] - 1.0) * (3.0 * psi[i] - 2.0);
            for (int m = 0; m < 6; ++m)
                for (int half = 0; half < 2; ++half) {  // phi_edge(closest, other)
                    const int cl = half ? ED[m][1] : ED[m][0], ot = half ? ED[m][0] : ED[m][1];
                    out[4 + 2 * m + half] = (9.0 / 2.0) * psi[cl] * psi[ot] * (3.0 * psi[cl] - 1.0);
                }
            for (int f = 0; f < 4; ++f) out[16 + f] = 27.0 * psi[FA[f][0]] * psi[FA[f][1]] * psi[FA[f][2]];
            break;
        }
        case FH_HEX20:  // hexahedron.rs:413-462
            for (int n = 0; n < 20; ++n) {
                const double al = HEX_SIGN[n][0], be = HEX_SIGN[n][1], ga = HEX_SIGN[n][2];
                const double g = (1.0 + al * xi[0]) * (1.0 + be * xi[1]) * (1.0 + ga * xi[2]);
                if (n < 8) out[n] = (1.0 / 8.0) * g * (al * xi[0] + be * xi[1] + ga * xi[2] - 2.0);
                else out[n] = (1.0 / 4.0) * (1.0 - (1.0 - al * al) * xi[0] * xi[0]) * (1.0 - (1.0 - be * be) * xi[1] * xi[1]) *
                              (1.0 - (1.0 - ga * ga) * xi[2] * xi[2]) * g;
            }
            break;
        case FH_QUAD9:  // quadrilateral.rs:247-277
            for (int n = 0; n < 9; ++n) out[n] = quad(QUAD9_SIGN[n][0], xi[0]) * quad(QUAD9_SIGN[n][1], xi[1]);
            break;
    }
}

struct ElemInfo { int d, n, ng, geom_kind; };
bool elem_info(int kind, ElemInfo& e) {
    switch (kind) {
        case FH_QUAD4: e = {2, 4, 4, FH_QUAD4}; return true;
        case FH_HEX8: e = {3, 8, 8, FH_HEX8}; return true;
        case FH_TET4: e = {3, 4, 4, FH_TET4}; return true;
        case FH_HEX27: e = {3, 27, 8, FH_HEX8}; return true;
        case FH_TRI3: e = {2, 3, 3, FH_TRI3}; return true;
        case FH_TET10: e = {3, 10, 4, FH_TET4}; return true;
        case FH_QUAD9: e = {2, 9, 4, FH_QUAD4}; return true;
        case FH_TRI6: e = {2, 6, 3, FH_TRI3}; return true;
        case FH_HEX20: e = {3, 20, 8, FH_HEX8}; return true;
        case FH_TET20: e = {3, 20, 4, FH_TET4}; return true;
        default: return false;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct fh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::string last_kernel;

    // mesh
    bool has_mesh = false, ragged = false;
    int elem_kind = -1;
    ElemInfo ei{};
    uint64_t N = 0, E = 0;
    DevBuf<double> verts;
    DevBuf<int> conn;           // flat node list
    DevBuf<unsigned> eoff, k2e; // ragged only
    uint64_t flat_len = 0;
    std::vector<uint64_t> h_eoff, h_nodes;  // host copy of the connectivity (colouring)
    bool has_host_conn = false;
    // operator / quadrature / u
    int op = -1;
    uint64_t sdim_ragged = 1;
    int nq = 0;
    DevBuf<double> qw, gref, ggeom, phiref, phigeom, qparams, u, rparams;
    DevBuf<unsigned> rule_map;
    bool has_rules = false;
    bool has_params = false, has_u = false;
    bool fast_ok = false;       // uniform parameters (or rules constant over their points, elem_par) and non-negative weights
    bool elem_par = false;      // compact table whose rules are constant over the points: pipelined kernel with per-slot data
    DevBuf<double> p_slotpar;
    bool has_slotpar = false;
    double uni_mu = 0.0, uni_lambda = 0.0;
    std::vector<double> h_points;
    // optional element mask: pattern from all elements, numerics from the active ones only
    bool has_mask = false;
    DevBuf<unsigned char> active;
    DevBuf<unsigned> active_list;   // indices of active elements (element-centric kernels)
    uint64_t num_active = 0;
    std::vector<unsigned char> h_active;
    DevBuf<unsigned> n2e_off_c, n2e_c;   // compute adjacency (active elements only)
    std::vector<unsigned> h_n2e_off_c;
    // pattern
    bool has_pattern = false;
    DevBuf<unsigned> noff, ncols, n2e_off, n2e;
    uint64_t nnz_nodes = 0;
    std::vector<unsigned> h_noff, h_n2e_off;  // host copies (gather block partition)
    // gather partition
    DevBuf<unsigned> blk_off, gt_elems, gt_ent;
    DevBuf<unsigned char> gt_pos;
    bool has_pos = false;
    // fixed-stride tables of the pipelined gather kernel
    DevBuf<int> p_conn, p_rec, p_elem;
    DevBuf<int> r_rec;          // row-owner kernel (rows_kernel.hpp): shared part of the records
    bool part_perm = false;     // the blocks were formed in a locality order of the nodes (row-owner Tet4 kernel only)
    bool part_rows_only = false;  // tables that only the row-owner Tet4 kernel can use (locality order and / or larger blocks)
    int rows_try = 0;           // block sizes tried for them: 0 = nine nodes / 256 entries, 1 = seven / 224, then the standard form
    DevBuf<uint4> r_lanes4;     //                                     lanes per position (Tet4)
    DevBuf<int> r_vconn;        //                                     unique vertices + slot words per position (Tet4)
    int r_rw = 0, r_ls = 256;
    bool has_rows = false;
    int p_rw = 0;
    int p_cs = 0, p_ms = 0, p_nbs = 0, p_jt = 1, p_us = 0;
    bool has_pipe = false;
    DevBuf<GatherHdr> gt_hdr;
    int nblk = 0, g_ub = 0, g_mb = 0, g_acc = 0, g_nb = 0, g_umax = 0;
    bool has_partition = false;
    // affine-element fast path (affine_kernel.hpp): per-element flags, reference blocks, and the sweep positions of the
    // node blocks all of whose elements are affine (the position-indexed tables above then cover the other blocks only)
    double affine_tol = 0x1p-46;
    bool has_aff = false;
    DevBuf<unsigned char> elem_aff;
    uint64_t num_aff = 0;
    DevBuf<double> ghat;            // [64][10] LinearElastic blocks | [64][6] Laplace blocks
    bool has_ghat = false;
    DevBuf<int> a_conn, a_elem;     // k_affine_rows (affine_rows.hip): per-slot connectivity (table build only), element ids
    DevBuf<int> a_vtab;             // ... vertex tables of the fused form (affine_rows_vertex_tables): [a_npos][a_nu + 32]
    int a_nu = 0;                   // padded length of their vertex lists (0: none -- the separate records kernel runs)
    DevBuf<uint2> a_lanes;          // lane records
    DevBuf<int4> a_hdr;             // position headers
    DevBuf<double> a_recs;          // element records (R or M), rewritten by every assembly
    int a_us = 0, a_npos = 0, a_ntab = 0, a_incomplete = 0;
    // general Hex8 row-owner kernel (hex8_rows.hip): lane tables and position records of the GENERAL positions (p_rec order)
    DevBuf<int4> h_hdr, h_pos;
    DevBuf<uint2> h_lanes;
    int h_ntab = 0, h_incomplete = 0;
    bool has_hrows = false;
    long long a_emin = 0, a_emax = -1;   // elements the affine positions of this partition refer to: the records kernel walks [a_emin, a_emax]
    unsigned max_row = 0;           // longest node-level row of the pattern (set by build_pattern)
    int npos_gen = 0;               // positions of the general tables (== nblk when no block is affine)
    bool aff_failed = false;        // the lane tables could not express an affine block of this mesh: general kernels only
    bool perm_failed = false;       // the locality order could not be used (no row-owner tables, or another kernel runs): natural order
    long long row_lo = 0, row_hi = -1;  // owner-computes node range (fh_set_row_range); row_hi < 0: all nodes
    // Second set of owner-computes tables (fh_assemble_matrix_rows_dev): the partition of another node range, swapped in for
    // the duration of that call.  struct_gen counts everything that invalidates a partition; the stash remembers the count
    // its tables were built at.
    unsigned long long struct_gen = 0;
    struct PartStash* rows_stash = nullptr;
    int status_slot = 0;                // DevStatus slot the kernels of the current call report to (1: the rows call)
    // Rule-set quadrature tables (fh_set_quadrature_rules: GeneralQuadratureTable, CompactQuadratureTable with different
    // point sets).  Rules with identical points and weights form a group; a group is staged as a uniform / compact table
    // with the element mask restricted to its elements, and the global assemblers walk the groups, accumulating.
    struct RuleSet {
        bool active = false;
        std::vector<uint64_t> offs;          // num_rules + 1: points of rule r are [offs[r], offs[r + 1])
        std::vector<double> w, pts, par;     // concatenated weights, points (x d), parameters (x 2; empty: none)
        std::vector<uint32_t> e2r;           // E
        std::vector<int> rule_group, rule_local;
        std::vector<std::vector<uint32_t>> groups;  // rules of each group
        int staged = -1;
    } rs;
    // Tuning / diagnostic switches: the FENRIS_HIP_* environment variables as they were when fh_create ran (read once; a
    // host that wants different settings sets them before creating the context -- see include/fenris_hip.h)
    std::unordered_map<std::string, std::string> env_vars;
    const char* env(const char* name) const {
        auto it = env_vars.find(name);
        return it == env_vars.end() ? nullptr : it->second.c_str();
    }
    int env_int(const char* name, int dflt) const {
        const char* v = env(name);
        return (v && *v) ? std::atoi(v) : dflt;
    }
    bool rs_staging = false;                 // the setters are being called by the group walk, not by the user
    std::vector<uint8_t> user_mask;          // fh_set_active_elements as the caller gave it
    bool user_has_mask = false;
    // colours
    bool has_colors = false;
    std::vector<uint64_t> color_offsets;
    std::vector<uint64_t> host_colors_offs, host_colors_labels;  // unfiltered colouring
    DevBuf<unsigned> labels;
    // status
    DevBuf<DevStatus> status;
    DevBuf<double> scratch;
    DevBuf<double> ke_dense;  // two-pass assembly of high-order elements: E dense element matrices
    DevBuf<double> fe_scratch;  // two-pass residual: E element vectors
    DevBuf<unsigned> src_n2e_off, src_n2e;   // node -> (element, local node) adjacency of a context without an operator (source vectors)
    unsigned long long src_adj_gen = ~0ull;
    DevBuf<double> scalar_partial;           // workgroup partials of the energy (kept: no allocation per call)
    VecTilesStore vt;                        // residual through element tiles (vector_tiles.hip)
    unsigned long long vt_gen = ~0ull;       // topo_gen the tiles were built for
    unsigned long long topo_gen = 0;         // counts fh_set_mesh calls (struct_gen also moves with vertex updates, masks, operators)
    bool vt_bad = false;
    DevBuf<unsigned char> tp_pos8;     // ... and the column slot per (entry, local node), 8 or 16 bit
    DevBuf<unsigned short> tp_pos16;
    bool has_tp_pos = false;
    DevBuf<unsigned long long> trace;
    bool defer_status = false;   // fh_assemble_vector_async_dev: the launches are only enqueued, fh_poll_status reports their errors
    bool keep_status = false;    // ... over a rule-set table: the status slot is reset once in front of the group walk, not per group

    int S() const {
        if (ragged) return (int)sdim_ragged;
        if (op < 0) return 0;
        return (op == FH_LAPLACE || op == FH_MASS_SCALAR) ? 1 : ei.d;
    }
    int fail(int code, const std::string& msg) { err = msg; return code; }
    int hip_fail(hipError_t e, const char* what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return FH_HIP_ERROR;
    }
};

// Everything build_partition produces (and the row range it was produced for), as a detachable unit.
#define FH_PARTITION_MEMBERS(X)                                                                                              \
    X(blk_off) X(gt_elems) X(gt_ent) X(gt_pos) X(has_pos) X(p_conn) X(p_rec) X(p_elem) X(r_rec) X(r_lanes4) X(r_vconn) X(r_rw) X(r_ls)  \
    X(has_rows) X(p_rw) X(p_cs) X(p_ms) X(p_nbs) X(p_jt) X(p_us) X(has_pipe) X(gt_hdr) X(nblk) X(g_ub) X(g_mb) X(g_acc)      \
    X(g_nb) X(g_umax) X(has_partition) X(a_conn) X(a_elem) X(a_vtab) X(a_nu) X(a_lanes) X(a_hdr) X(a_us) X(a_npos) X(a_ntab) X(a_incomplete) X(a_emin) X(a_emax) X(npos_gen)       \
    X(h_hdr) X(h_pos) X(h_lanes) X(h_ntab) X(h_incomplete) X(has_hrows) X(aff_failed) X(row_lo) X(row_hi) X(p_slotpar) X(has_slotpar) X(part_perm) X(part_rows_only) X(rows_try) X(perm_failed)
struct PartStash {
#define X(name) decltype(fh_ctx::name) name{};
    FH_PARTITION_MEMBERS(X)
#undef X
    unsigned long long built_gen = ~0ull;
    PartStash() { row_lo = 0; row_hi = -1; r_ls = 256; p_jt = 1; }
};
template <class T> static void part_swap(DevBuf<T>& a, DevBuf<T>& b) { std::swap(a.p, b.p); std::swap(a.n, b.n); }
template <class T> static void part_swap(T& a, T& b) { std::swap(a, b); }
static void swap_partition(fh_ctx* c, PartStash& st) {
#define X(name) part_swap(c->name, st.name);
    FH_PARTITION_MEMBERS(X)
#undef X
}

// Every entry point that touches the device runs on the context's device and leaves the calling thread's current device as it
// found it (several contexts on different GPUs in one process; torch's current device is the thread's too).
struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != dev) { prev = cur; (void)hipSetDevice(dev); }
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};

#define HIP_TRY(ctx, expr)                                        \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return (ctx)->hip_fail(_e, #expr);  \
    } while (0)

namespace {

int grid_for(long long n, int block, int cap = 256 * 32) {
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
    c->h_noff.resize(N + 1);
    c->h_n2e_off.resize(N + 1);
    HIP_TRY(c, hipMemcpyAsync(c->h_noff.data(), c->noff.p, sizeof(unsigned) * (N + 1), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_n2e_off.data(), c->n2e_off.p, sizeof(unsigned) * (N + 1), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
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
    bool once = N > 0 && !c->ragged && max_cand <= 128ull && !c->env("FENRIS_HIP_PATTERN_TWO_PASSES");
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
    c->has_partition = false; ++c->struct_gen; c->has_tp_pos = false;
    return build_compute_adjacency(c);
}

// ---------------------------------------------------------------------------------- kernel dispatch
template <int EK, int OP>
int launch_matrix(fh_ctx* c, KArgs& a, int mode, size_t lds_bytes, int grid) {
    if (grid <= 0) return FH_OK;   // nothing to do (an element mask without an active element): a launch of zero workgroups is an error
    hipStream_t st = c->stream;
#define FH_LAUNCH(M)                                                                                              \
    do {                                                                                                          \
        auto kern = k_assemble_matrix<EK, OP, M>;                                                                 \
        if (lds_bytes > 48 * 1024)                                                                                \
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)lds_bytes));                                                       \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, st, a);                                        \
    } while (0)
    switch (mode) {
        case MODE_ATOMIC: FH_LAUNCH(MODE_ATOMIC); break;
        case MODE_COLORED: FH_LAUNCH(MODE_COLORED); break;
        case MODE_GATHER: FH_LAUNCH(MODE_GATHER); break;
        case MODE_DUMP: FH_LAUNCH(MODE_DUMP); break;
        default: return c->fail(FH_BAD_ARGUMENT, "bad scatter mode");
    }
#undef FH_LAUNCH
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

template <int EK, int OP>
size_t layout_bytes(int what, int nq, int ub, int acc, int nb, bool gather, int mb, int fast, int nc_row) {
    switch (what) {
        case WHAT_MATRIX: return make_layout<EK, OP, WHAT_MATRIX>(nq, ub, acc, nb, gather, mb, fast, 0, nc_row).bytes();
        case WHAT_VECTOR: return make_layout<EK, OP, WHAT_VECTOR>(nq, ub, acc, nb, gather, mb).bytes();
        default: return make_layout<EK, OP, WHAT_SCALAR>(nq, ub, acc, nb, gather, mb).bytes();
    }
}

// dispatch over (element kind, operator kind) -> template instantiation
#define FH_FOR_ELEM_OP(EKV, OPV, CALL)                                             \
    switch (EKV) {                                                                 \
        case FH_QUAD4: FH_FOR_OP(FH_QUAD4, OPV, CALL); break;                      \
        case FH_HEX8: FH_FOR_OP(FH_HEX8, OPV, CALL); break;                        \
        case FH_TET4: FH_FOR_OP(FH_TET4, OPV, CALL); break;                        \
        case FH_HEX27: FH_FOR_OP(FH_HEX27, OPV, CALL); break;                      \
        case FH_TRI3: FH_FOR_OP(FH_TRI3, OPV, CALL); break;                        \
        case FH_TET10: FH_FOR_OP(FH_TET10, OPV, CALL); break;                      \
        case FH_QUAD9: FH_FOR_OP(FH_QUAD9, OPV, CALL); break;                      \
        case FH_TRI6: FH_FOR_OP(FH_TRI6, OPV, CALL); break;                        \
        case FH_HEX20: FH_FOR_OP(FH_HEX20, OPV, CALL); break;                      \
        case FH_TET20: FH_FOR_OP(FH_TET20, OPV, CALL); break;                      \
        default: break;                                                            \
    }
#define FH_FOR_OP(EKC, OPV, CALL)                                   \
    switch (OPV) {                                                  \
        case FH_LAPLACE: CALL(EKC, FH_LAPLACE); break;              \
        case FH_LINEAR_ELASTIC: CALL(EKC, FH_LINEAR_ELASTIC); break;\
        case FH_NEO_HOOKEAN: CALL(EKC, FH_NEO_HOOKEAN); break;      \
        case FH_STVK: CALL(EKC, FH_STVK); break;                    \
        case FH_MASS_SCALAR: CALL(EKC, FH_MASS_SCALAR); break;      \
        case FH_MASS_VECTOR: CALL(EKC, FH_MASS_VECTOR); break;      \
        default: break;                                             \
    }

size_t layout_bytes_dyn(int ek, int op, int what, int nq, int ub, int acc, int nb, bool gather, int mb = 0, int fast = 0, int nc_row = 0) {
    size_t r = 0;
#define CALL(EKC, OPC) r = layout_bytes<EKC, OPC>(what, nq, ub, acc, nb, gather, mb, fast, nc_row)
    FH_FOR_ELEM_OP(ek, op, CALL)
#undef CALL
    return r;
}

constexpr size_t LDS_TARGET = 64 * 1024;   // two workgroups per CU
constexpr size_t LDS_LIMIT = 160 * 1024;   // hardware limit per workgroup

int check_ready(fh_ctx* c, const char* who, bool need_pattern) {
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, std::string(who) + ": no finite element mesh set");
    if (c->op < 0) return c->fail(FH_INVALID_STATE, std::string(who) + ": no operator set");
    if (c->nq <= 0) return c->fail(FH_INVALID_STATE, std::string(who) + ": no quadrature table set");
    if (c->op != FH_LAPLACE && !c->has_params)
        return c->fail(FH_INVALID_STATE, std::string(who) + ": operator needs per-point parameters (mu, lambda)");
    if (need_pattern && !c->has_pattern) return c->fail(FH_INVALID_STATE, std::string(who) + ": call fh_pattern first");
    return FH_OK;
}

// the pre-scaled-gradient ("fast") form of every kernel but the pipelined gather needs ONE uniform parameter pair;
// with a compact table only the pipelined kernel knows per-element data (fh_ctx::elem_par)
static bool generic_fast(const fh_ctx* c) { return c->fast_ok && (!c->has_rules || c->op == FH_LAPLACE); }

void fill_common(fh_ctx* c, KArgs& a) {
    std::memset(&a, 0, sizeof a);
    a.verts = c->verts.p;
    a.conn = c->conn.p;
    a.num_elements = (long long)c->E;
    a.num_nodes = (int)c->N;
    a.nq = c->nq;
    a.qw = c->qw.p;
    a.gref = c->gref.p;
    a.ggeom = c->ggeom.p;
    a.phiref = c->phiref.p;
    a.qparams = c->has_params ? c->qparams.p : nullptr;
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
int choose_epb(fh_ctx* c, int what) {
    int best = 1;
    for (int epb = 1; epb <= 64; ++epb) {
        const size_t b = layout_bytes_dyn(c->elem_kind, c->op, what, c->nq, epb, 0, 0, false, 0, generic_fast(c));
        if (b <= LDS_TARGET) best = epb; else break;
    }
    return best;
}

// Lane tables of the row-owner kernels (k_affine_rows, k_hex8_rows) for the positions described by the pipelined kernel's records
// `rec`: one record of 256 lanes per position (affine_rows_build), positions with identical records share one table (hashed on the
// device, merged here, verified on the device), the table id goes into every header.  `bad`: some block cannot be expressed.
static int build_lane_tables(fh_ctx* c, const int* rec, int us, int ms, int nb_target, int npos, int S, const int* conn, const int* elem,
                             DevBuf<int4>& hdr, DevBuf<uint2>& lanes, int& ntab_out, int& incomplete_out, bool& bad_out, const char* what,
                             int mirror = 0) {
    DevBuf<int> st;
    HIP_TRY(c, st.alloc(2));
    HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
    HIP_TRY(c, hdr.alloc((size_t)npos));
    // Hash-only build (round 4): the 256 records of a position are formed in LDS, hashed twice (128 bits) and dropped; the records of the
    // first position of every distinct table are formed once more into the compact tables.  Writing all of them (2 KB x 1.46 M positions
    // = 3 GB on the 216^3 mesh) cost an allocation of 40 - 120 ms.  FENRIS_HIP_LANE_TABLES_FULL keeps the full form (its compaction
    // compares every position with its table); two positions whose first hashes agree and whose second ones differ send the build there too.
    bool full = c->env("FENRIS_HIP_LANE_TABLES_FULL") != nullptr || c->env("FENRIS_HIP_NO_LANE_DEDUPE") != nullptr || npos >= (1 << 23);
    DevBuf<uint2> lanes_full;
    DevBuf<unsigned long long> hash_d;
    HIP_TRY(c, hash_d.alloc((size_t)npos * 2));
    std::vector<unsigned long long> hash_h((size_t)npos * 2);
    int bad = 0;
    if (!full) {
        HIP_TRY(c, affine_rows_build(c->stream, rec, c->p_rw, us, ms, nb_target, npos, S, c->ncols.p, conn, c->p_cs, elem, hdr.p, (uint2*)nullptr, st.p,
                                     hash_d.p, mirror, hash_d.p + npos));
        HIP_TRY(c, hipMemcpyAsync(&bad, st.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(hash_h.data(), hash_d.p, sizeof(unsigned long long) * (size_t)npos * 2, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        bad_out = bad != 0;
        if (bad) return FH_OK;
        std::vector<int> ids((size_t)npos), first;
        std::unordered_map<unsigned long long, int> seen;
        seen.reserve(1024);
        bool collision = false;
        for (int p = 0; p < npos && !collision; ++p) {
            auto it = seen.find(hash_h[p]);
            if (it == seen.end()) {
                it = seen.emplace(hash_h[p], (int)first.size()).first;
                first.push_back(p);
            } else if (hash_h[(size_t)npos + first[it->second]] != hash_h[(size_t)npos + p]) {
                collision = true;
            }
            ids[p] = it->second;
        }
        if (!collision) {
            const int ntab = (int)first.size();
            DevBuf<int> ids_d, first_d;
            HIP_TRY(c, ids_d.alloc((size_t)npos));
            HIP_TRY(c, first_d.alloc((size_t)ntab));
            HIP_TRY(c, lanes.alloc((size_t)ntab * 256));
            HIP_TRY(c, hipMemcpyAsync(ids_d.p, ids.data(), sizeof(int) * (size_t)npos, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(first_d.p, first.data(), sizeof(int) * (size_t)ntab, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
            HIP_TRY(c, affine_rows_tables(c->stream, rec, c->p_rw, us, ms, nb_target, npos, S, c->ncols.p, conn, c->p_cs, elem, mirror, ids_d.p, first_d.p,
                                          ntab, lanes.p, hdr.p, st.p));
            int mismatch[2] = {0, 0};   // [1]: some position has a block without an owner lane
            HIP_TRY(c, hipMemcpyAsync(mismatch, st.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            ntab_out = ntab;
            incomplete_out = mismatch[1];
            if (c->env("FENRIS_HIP_VERBOSE")) {
                long long changes = 0;
                for (int p = 1; p < npos; ++p) changes += ids[p] != ids[p - 1];
                std::fprintf(stderr, "[fenris_hip] %s: %d positions share %d lane tables, %lld changes of table along the sweep%s\n", what,
                             npos, ntab_out, changes, incomplete_out ? ", some position has a block without an owner" : "");
            }
            return FH_OK;
        }
        full = true;
        HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
    }
    HIP_TRY(c, lanes_full.alloc((size_t)npos * 256));
    HIP_TRY(c, affine_rows_build(c->stream, rec, c->p_rw, us, ms, nb_target, npos, S, c->ncols.p, conn, c->p_cs, elem, hdr.p, lanes_full.p, st.p,
                                 hash_d.p, mirror));
    HIP_TRY(c, hipMemcpyAsync(&bad, st.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hash_h.data(), hash_d.p, sizeof(unsigned long long) * (size_t)npos, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    bad_out = bad != 0;
    if (bad) return FH_OK;
    // positions with identical lane records (the interior of a structured mesh) share one table: the kernel
    // skips the fetch when the table does not change, and what it fetches stays in the caches
    std::vector<int> ids((size_t)npos), first;
    auto dedupe = [&](bool identity) {
        first.clear();
        if (identity) {
            first.resize((size_t)npos);
            for (int p = 0; p < npos; ++p) { ids[p] = p; first[p] = p; }
            return;
        }
        std::unordered_map<unsigned long long, int> seen;
        seen.reserve(1024);
        for (int p = 0; p < npos; ++p) {
            auto it = seen.find(hash_h[p]);
            if (it == seen.end()) {
                it = seen.emplace(hash_h[p], (int)first.size()).first;
                first.push_back(p);
            }
            ids[p] = it->second;
        }
    };
    dedupe(c->env("FENRIS_HIP_NO_LANE_DEDUPE") != nullptr || npos >= (1 << 23));  // the id has 23 bits
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int ntab = (int)first.size();
        DevBuf<int> ids_d, first_d;
        HIP_TRY(c, ids_d.alloc((size_t)npos));
        HIP_TRY(c, first_d.alloc((size_t)ntab));
        HIP_TRY(c, lanes.alloc((size_t)ntab * 256));
        HIP_TRY(c, hipMemcpyAsync(ids_d.p, ids.data(), sizeof(int) * (size_t)npos, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(first_d.p, first.data(), sizeof(int) * (size_t)ntab, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(st.p, 0, 2 * sizeof(int), c->stream));
        HIP_TRY(c, affine_rows_compact(c->stream, lanes_full.p, ids_d.p, first_d.p, npos, ntab, lanes.p, hdr.p, st.p));
        int mismatch[2] = {0, 0};   // [1]: some position has a block without an owner lane
        HIP_TRY(c, hipMemcpyAsync(mismatch, st.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        ntab_out = ntab;
        incomplete_out = mismatch[1];
        if (!mismatch[0]) break;
        dedupe(true);  // a hash collision: every position keeps its own table
    }
    if (c->env("FENRIS_HIP_VERBOSE")) {
        long long changes = 0;   // positions whose table differs from their predecessor's in the sweep: each is a 2 KB fetch
        for (int p = 1; p < npos; ++p) changes += ids[p] != ids[p - 1];
        std::fprintf(stderr, "[fenris_hip] %s: %d positions share %d lane tables, %lld changes of table along the sweep%s\n", what,
                     npos, ntab_out, changes, incomplete_out ? ", some position has a block without an owner" : "");
    }
    return FH_OK;
}

// greedy partition of the node range into owner blocks (gather mode)

int build_partition(fh_ctx* c) {
    if (c->has_partition) return FH_OK;
    // FENRIS_HIP_VERBOSE: wall time of the stages of this set-up (stream drained at every mark)
    auto t_last = std::chrono::steady_clock::now();
    const bool vt = c->env("FENRIS_HIP_VERBOSE") != nullptr;
    auto mark = [&](const char* what) {
        if (!vt) return;
        (void)hipStreamSynchronize(c->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[fenris_hip] set-up: %-34s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    { const int rc_h = host_offsets(c); if (rc_h) return rc_h; }
    mark("host copies of the offsets");
    // adjacency that drives the numerics: all elements, or only the active ones when a mask is set
    const std::vector<unsigned>& adj_off_h = c->has_mask ? c->h_n2e_off_c : c->h_n2e_off;
    const unsigned* adj_off_d = c->has_mask ? c->n2e_off_c.p : c->n2e_off.p;
    const unsigned* adj_d = c->has_mask ? c->n2e_c.p : c->n2e.p;
    const int S = c->S();
    const int N = (int)c->N;
    const unsigned max_row = c->max_row;
    // The owner blocks are ranges of consecutive nodes.  On a mesh whose numbering has no locality (consecutive nodes share no
    // element: every node of a block brings its own ~24 tetrahedra, C3: 128 slots and 11 kB of vertex gathers for 5 nodes) the
    // row-owner Tet4 kernel -- whose lanes store every block of a row straight to its place, so that the rows of a block
    // need not be neighbours in memory -- gets its blocks from a locality order instead: nodes sorted by the Morton key of
    // their coordinates, the pattern rows and the node -> element adjacency permuted alike (contents unchanged: real node
    // and element ids), the real first entry of every row handed to the kernel (r_rec).  Everything below then works on
    // positions in that order; nothing else in the context sees it.
    const unsigned* noff_d = c->noff.p;
    const unsigned* ncols_d = c->ncols.p;
    const std::vector<unsigned>* h_noff_p = &c->h_noff;
    const std::vector<unsigned>* adj_off_hp = &adj_off_h;
    DevBuf<unsigned> v2r_d, noff_v, ncols_v, adj_off_v, adj_v;
    DevBuf<int> r2v_d;
    std::vector<unsigned> h_noff_v, adj_off_hv;
    c->part_perm = false;
    const bool perm_cand = c->elem_kind == FH_TET4 && (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC) && c->row_hi < 0 &&
                           !c->perm_failed && !c->has_rules && c->fast_ok && N > 64 && !c->env("FENRIS_HIP_NO_ROWS") &&
                           !c->env("FENRIS_HIP_NO_NODE_ORDER") && !c->env("FENRIS_HIP_TRACE");
    if (perm_cand) {
        // how local is the numbering?  fraction of nodes that share an element with their successor
        DevBuf<unsigned char> link_d;
        HIP_TRY(c, link_d.alloc((size_t)N + 1));
        hipLaunchKernelGGL(k_linked_to_next, dim3((N + 255) / 256), dim3(256), 0, c->stream, adj_off_d, adj_d, c->ei.n, N, link_d.p);
        std::vector<unsigned char> lk((size_t)N);
        HIP_TRY(c, hipMemcpyAsync(lk.data(), link_d.p, (size_t)N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        long long linked = 0;
        for (int i = 0; i < N; ++i) linked += lk[i];
        const bool force = c->env_int("FENRIS_HIP_NODE_ORDER", 0) != 0;
        if (force || linked * 2 < (long long)N) {
            const int D = c->ei.d;
            std::vector<double> hv((size_t)N * D);
            HIP_TRY(c, hipMemcpyAsync(hv.data(), c->verts.p, sizeof(double) * hv.size(), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
            for (int k = 0; k < D; ++k) { lo[k] = hv[k]; hi[k] = hv[k]; }
            for (size_t i = 0; i < (size_t)N; ++i)
                for (int k = 0; k < D; ++k) { lo[k] = std::min(lo[k], hv[i * D + k]); hi[k] = std::max(hi[k], hv[i * D + k]); }
            for (int k = 0; k < D; ++k) sc[k] = (hi[k] > lo[k]) ? 2097151.0 / (hi[k] - lo[k]) : 0.0;
            DevBuf<unsigned long long> keys, keys_s;
            DevBuf<unsigned> ids;
            HIP_TRY(c, keys.alloc((size_t)N));
            HIP_TRY(c, keys_s.alloc((size_t)N));
            HIP_TRY(c, ids.alloc((size_t)N));
            HIP_TRY(c, v2r_d.alloc((size_t)N));
            hipLaunchKernelGGL(k_morton_keys, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->verts.p, N, D, lo[0], lo[1], lo[2], sc[0],
                               sc[1], sc[2], keys.p, ids.p);
            size_t tb = 0;
            HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys.p, keys_s.p, ids.p, v2r_d.p, N, 0, 64, c->stream));
            DevBuf<char> tmp;
            HIP_TRY(c, tmp.alloc(tb + 16));
            HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, keys.p, keys_s.p, ids.p, v2r_d.p, N, 0, 64, c->stream));
            HIP_TRY(c, r2v_d.alloc((size_t)N));
            hipLaunchKernelGGL(k_invert_perm, dim3((N + 255) / 256), dim3(256), 0, c->stream, v2r_d.p, N, r2v_d.p);
            // rows of the pattern and of the adjacency in that order
            auto permute_rows = [&](const unsigned* off_src, const unsigned* src, size_t total, DevBuf<unsigned>& off_dst, DevBuf<unsigned>& dst,
                                    std::vector<unsigned>& off_h) -> int {
                DevBuf<unsigned> len;
                HIP_TRY(c, len.alloc((size_t)N + 1));
                HIP_TRY(c, off_dst.alloc((size_t)N + 1));
                HIP_TRY(c, dst.alloc(total + 1));
                hipLaunchKernelGGL(k_perm_row_lengths, dim3((N + 256) / 256), dim3(256), 0, c->stream, off_src, v2r_d.p, N, len.p);
                size_t sb = 0;
                HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, sb, len.p, off_dst.p, N + 1, c->stream));
                DevBuf<char> t2;
                HIP_TRY(c, t2.alloc(sb + 16));
                HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(t2.p, sb, len.p, off_dst.p, N + 1, c->stream));
                hipLaunchKernelGGL(k_perm_copy_rows, dim3((N + 255) / 256), dim3(256), 0, c->stream, off_src, src, v2r_d.p, off_dst.p, dst.p, N);
                HIP_TRY(c, hipGetLastError());
                off_h.resize((size_t)N + 1);
                HIP_TRY(c, hipMemcpyAsync(off_h.data(), off_dst.p, sizeof(unsigned) * ((size_t)N + 1), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));  // len / t2 are released on return
                return FH_OK;
            };
            int rp = permute_rows(c->noff.p, c->ncols.p, (size_t)c->h_noff[N], noff_v, ncols_v, h_noff_v);
            if (rp) return rp;
            rp = permute_rows(adj_off_d, adj_d, (size_t)adj_off_h[N], adj_off_v, adj_v, adj_off_hv);
            if (rp) return rp;
            noff_d = noff_v.p;
            ncols_d = ncols_v.p;
            adj_off_d = adj_off_v.p;
            adj_d = adj_v.p;
            h_noff_p = &h_noff_v;
            adj_off_hp = &adj_off_hv;
            c->part_perm = true;
            if (c->env("FENRIS_HIP_VERBOSE"))
                std::fprintf(stderr, "[fenris_hip] node numbering without locality (%.1f %% of the nodes share an element with their successor): "
                                     "owner blocks formed in Morton order\n", 100.0 * (double)linked / (double)N);
        }
    }
    const std::vector<unsigned>& h_noff = *h_noff_p;
    const std::vector<unsigned>& adj_off_hh = *adj_off_hp;
    // nodes per block (tunable), entry capacity per batch, accumulator budget
    // Hex8 meshes with affine elements: 36 row lanes per node in k_affine_rows, seven nodes per block also for S = 1
    const bool aff_cand = c->elem_kind == FH_HEX8 && c->has_aff && c->num_aff > 0 && !c->aff_failed && c->affine_tol > 0.0 &&
                          (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC || (c->op == FH_MASS_SCALAR && c->has_params));
    const bool rows_special = perm_cand;   // tables for the row-owner Tet4 kernel alone: larger blocks (below)
    // Hex8 Laplace / LinearElastic without a mask: the general positions run on k_hex8_rows (36 row lanes per node as well)
    const bool hrows_cand = c->elem_kind == FH_HEX8 && (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC) && !c->has_rules &&
                            !c->env("FENRIS_HIP_NO_HEX8_ROWS");
    const int nb_target = std::max(1, std::min(64, c->env_int("FENRIS_HIP_GATHER_NB", rows_special ? (c->rows_try == 0 ? 9 : 7)
                                                                                                     : (S == 1 && !aff_cand && !hrows_cand) ? 8 : 7)));  // < 256: packed in 8 bits
    // Tables for the row-owner Tet4 kernel alone may hold more entries per block than the pipelined kernel's lane mapping takes
    // and more nodes (the lane word has four bits for the node): nine nodes / 256 entries first (C3: 98 k positions of ~170 lanes
    // instead of 171 k of ~90, 0.80 -> 0.64 ms), seven / 224 when that cannot be expressed (0.67 ms), then the standard form
    c->part_rows_only = rows_special;
    const int mb = std::max(16, std::min(1024, c->env_int("FENRIS_HIP_GATHER_MB", rows_special ? (c->rows_try == 0 ? 256 : 224) : 128)));
    const size_t lds_target = (size_t)c->env_int("FENRIS_HIP_GATHER_LDS_KB", 52) * 1024;
    // accumulators: nb_target typical rows, but at least the largest single row block
    long long sum_rows = 0;
    for (int i = 0; i < N; ++i) sum_rows += h_noff[i + 1] - h_noff[i];
    const int avg_row = N ? (int)((sum_rows + N - 1) / N) : 1;
    int acc = S * S * std::max<int>((int)max_row, std::min<int>(nb_target * (avg_row + avg_row / 4 + 1), 8192 / (S * S)));
    std::vector<unsigned> blk;
    // owner-computes covers the nodes [n_lo, n_hi): everything, or the range of fh_set_row_range
    const int n_lo = (c->row_hi < 0) ? 0 : (int)std::min<long long>(c->row_lo, N);
    const int n_hi = (c->row_hi < 0) ? N : (int)std::min<long long>(c->row_hi, N);
    blk.push_back((unsigned)n_lo);
    // Blocks are aligned to RUNS of consecutive nodes that share an element with their successor (the grid lines of
    // a structured numbering): a run of L >= nb_target nodes is cut into ceil(L / nb_target) blocks of balanced size,
    // so that every line of a structured mesh is cut at the same places and consecutive blocks of a sweep chain
    // share exactly the elements between two lines.  Short runs (unstructured numberings) are merged greedily.
    std::vector<unsigned char> link((size_t)N + 1, 1);
    if (N > 0 && !c->env("FENRIS_HIP_NO_ALIGN")) {
        DevBuf<unsigned char> link_d;
        HIP_TRY(c, link_d.alloc((size_t)N + 1));
        hipLaunchKernelGGL(k_linked_to_next, dim3((N + 255) / 256), dim3(256), 0, c->stream, adj_off_d, adj_d, c->ei.n, N, link_d.p);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(link.data(), link_d.p, (size_t)N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    mark("row sums, link flags");
    auto fits = [&](int a, int b) {  // nodes [a, b) within the accumulator and entry budgets
        return S * S * ((long long)h_noff[b] - h_noff[a]) <= acc && (long long)adj_off_hh[b] - adj_off_hh[a] <= mb;
    };
    auto cut_greedy = [&](int a, int b) {  // blocks of up to nb_target nodes, shrunk where the budgets demand it
        while (a < b) {
            int e = std::min(b, a + nb_target);
            while (e > a + 1 && !fits(a, e)) --e;
            blk.push_back((unsigned)e);
            a = e;
        }
    };
    int i0 = n_lo;
    while (i0 < n_hi) {
        int r1 = i0 + 1;  // end of the run that starts at i0
        while (r1 < n_hi && link[r1 - 1]) ++r1;
        const int L = r1 - i0;
        if (L >= nb_target) {
            const int k = (L + nb_target - 1) / nb_target;
            bool ok = true;
            std::vector<int> ends;
            for (int j = 1; j <= k && ok; ++j) {
                const int e = i0 + (int)((long long)L * j / k);
                ok = fits(ends.empty() ? i0 : ends.back(), e);
                ends.push_back(e);
            }
            if (ok) for (int e : ends) blk.push_back((unsigned)e);
            else cut_greedy(i0, r1);
            i0 = r1;
        } else {
            // short runs (unstructured numbering): the whole stretch up to the next long run is cut greedily
            int e = r1;
            while (e < n_hi) {
                int r2 = e + 1;
                while (r2 < n_hi && link[r2 - 1]) ++r2;
                if (r2 - e >= nb_target) break;
                e = r2;
            }
            cut_greedy(i0, e);
            i0 = e;
        }
    }
    mark("cutting the node range (host)");
    c->nblk = (int)blk.size() - 1;
    if (c->nblk <= 0) {  // empty row range: nothing to build, nothing to launch
        c->nblk = 0;
        c->has_pipe = false;
        c->has_rows = false;
        c->has_slotpar = false;
        c->a_npos = 0;
        c->npos_gen = 0;
        c->has_partition = true;
        return FH_OK;
    }
    {   // tighten the accumulator budget to the largest block actually formed
        long long mx = 1;
        for (size_t b = 0; b + 1 < blk.size(); ++b) mx = std::max<long long>(mx, (long long)h_noff[blk[b + 1]] - h_noff[blk[b]]);
        acc = (int)(S * S * mx);
    }
    HIP_TRY(c, c->blk_off.alloc(blk.size()));
    HIP_TRY(c, hipMemcpyAsync(c->blk_off.p, blk.data(), sizeof(unsigned) * blk.size(), hipMemcpyHostToDevice, c->stream));
    // block tables: unique element lists and packed entries (built once per pattern/partition)
    {
        unsigned max_m = 0;
        for (size_t b = 0; b + 1 < blk.size(); ++b)
            max_m = std::max(max_m, adj_off_hh[blk[b + 1]] - adj_off_hh[blk[b]]);
        if (max_m >= 65536) return c->fail(FH_UNSUPPORTED, "gather mode: a node block has more than 65535 adjacent entries");
        const size_t tb = sizeof(int) * 3 * (size_t)std::max(1u, max_m);
        if (tb > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "gather mode: node valence too large for the table builder");
        const int nblk = c->nblk;
        DevBuf<unsigned> counts, uoff;
        HIP_TRY(c, c->gt_hdr.alloc((size_t)nblk + 1));
        HIP_TRY(c, counts.alloc((size_t)nblk + 1));
        HIP_TRY(c, uoff.alloc((size_t)nblk + 1));
        HIP_TRY(c, c->gt_ent.alloc((size_t)c->flat_len + 1));
        auto k0 = k_build_gather_tables<0>;
        auto k1 = k_build_gather_tables<1>;
        if (tb > 48 * 1024) {
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb));
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb));
        }
        c->has_pos = max_row < 256 && !c->env("FENRIS_HIP_NO_POS");
        if (c->has_pos) HIP_TRY(c, c->gt_pos.alloc((size_t)c->flat_len * c->ei.n + 4));
        hipLaunchKernelGGL(k0, dim3(nblk), dim3(256), tb, c->stream, c->blk_off.p, noff_d, adj_off_d, adj_d, c->ei.n,
                           c->gt_hdr.p, (const unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                           (const unsigned*)nullptr, (unsigned char*)nullptr);
        hipLaunchKernelGGL(k_hdr_counts, dim3((nblk + 256) / 256), dim3(256), 0, c->stream, c->gt_hdr.p, nblk, counts.p);
        size_t tmpb = 0;
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tmpb, counts.p, uoff.p, nblk + 1, c->stream));
        DevBuf<char> tmp;
        HIP_TRY(c, tmp.alloc(tmpb + 16));
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(tmp.p, tmpb, counts.p, uoff.p, nblk + 1, c->stream));
        unsigned total_u = 0;
        HIP_TRY(c, hipMemcpyAsync(&total_u, uoff.p + nblk, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, c->gt_elems.alloc((size_t)total_u + 1));
        hipLaunchKernelGGL(k1, dim3(nblk), dim3(256), tb, c->stream, c->blk_off.p, noff_d, adj_off_d, adj_d, c->ei.n,
                           c->gt_hdr.p, uoff.p, c->gt_elems.p, c->gt_ent.p, c->conn.p, ncols_d,
                           c->has_pos ? c->gt_pos.p : (unsigned char*)nullptr);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    mark("block tables (k_build_gather_tables)");
    // staging capacity: all unique elements of the largest block if that fits the LDS budget
    std::vector<GatherHdr> hh((size_t)std::max(c->nblk, 1));
    if (c->nblk) HIP_TRY(c, hipMemcpy(hh.data(), c->gt_hdr.p, sizeof(GatherHdr) * (size_t)c->nblk, hipMemcpyDeviceToHost));
    int umax = 1, mmax = 1;
    for (int b = 0; b < c->nblk; ++b) { umax = std::max(umax, hh[b].U); mmax = std::max(mmax, hh[b].m); }
    int ub = 0;
    if (layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, umax, acc, 64, true, mb, c->fast_ok) <= lds_target) {
        ub = umax;
    } else {
        for (int t = 1; t <= umax; ++t) {
            const size_t b = layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, t, acc, 64, true, mb, c->fast_ok);
            if (b <= lds_target) ub = t; else break;
        }
    }
    if (ub == 0) {
        ub = 1;
        if (layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, 1, acc, 64, true, mb, c->fast_ok) > LDS_LIMIT)
            return c->fail(FH_UNSUPPORTED, "gather mode: a row block does not fit in LDS; use FH_SCATTER_ATOMIC");
    }
    if (c->env("FENRIS_HIP_VERBOSE"))
        std::fprintf(stderr, "[fenris_hip] gather partition: nblk=%d nb=%d umax=%d mmax=%d acc=%d ub=%d lds=%zu B\n", c->nblk,
                     nb_target, umax, mmax, acc, ub,
                     layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, ub, acc, 64, true, mb, c->fast_ok));
    // position-indexed tables for the pipelined kernel (elements with few geometry nodes, pos table present)
    c->has_pipe = false;
    c->has_rows = false;
    c->a_npos = 0;
    c->npos_gen = c->nblk;
    if (c->has_pos && !c->env("FENRIS_HIP_NO_PIPE") && c->ei.n == c->ei.ng && c->ei.n <= 8 && c->nblk > 0) {
        const int n = c->ei.n;
        const int ms = (mmax + 3) / 4 * 4;
        const int us = (umax + 3) / 4 * 4;
        // local nodes per lane in the pipelined kernel's phase C
        int jt = c->env_int("FENRIS_HIP_PIPE_JT", (n % 2 == 0) ? 2 : n);
        if (jt != 1 && jt != 2 && jt != 4 && jt != n) jt = 1;
        if (n % jt != 0) jt = 1;
        c->p_jt = jt;
        if (us * c->ei.ng <= (rows_special ? 1024 : 512) && us <= 252 && ms <= 256 && (rows_special || (ms * (n / jt) <= 256 && ms * n / 4 <= 256)) && ms <= mb &&
            nb_target <= 254 && pipe_record_words(us, ms, n, nb_target) <= 512 &&
            (c->fast_ok || (c->op == FH_MASS_SCALAR && c->elem_kind == FH_HEX8))) {   // (the mass tables take densities that differ from point to point)
            const int nblk = c->nblk;
            mark("headers to the host, staging sizes");
            // Block classes: 1 = every adjacent element is affine, the block runs on k_affine_rows; 0 = general kernels.
            // Chains never mix classes, so each class gets its own sweep order and its own position-indexed tables.
            std::vector<unsigned char> cls((size_t)nblk, 0);
            DevBuf<unsigned char> cls_d;
            const bool want_aff = c->elem_kind == FH_HEX8 && c->has_aff && c->has_ghat && c->num_aff > 0 && !c->has_rules &&
                                  !c->aff_failed && c->affine_tol > 0.0 &&
                                  (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC || (c->op == FH_MASS_SCALAR && c->has_params)) &&
                                  us <= 32 && nb_target <= 8 && !c->env("FENRIS_HIP_NO_AFFINE");
            if (want_aff) {
                HIP_TRY(c, cls_d.alloc((size_t)nblk));
                hipLaunchKernelGGL(k_block_class, dim3((nblk + 255) / 256), dim3(256), 0, c->stream, c->gt_hdr.p, c->gt_elems.p,
                                   c->elem_aff.p, nblk, 32, cls_d.p);
                HIP_TRY(c, hipGetLastError());
                HIP_TRY(c, hipMemcpyAsync(cls.data(), cls_d.p, (size_t)nblk, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            }
            mark("block classes");
            // sweep order: chains of blocks whose consecutive members share elements (their staged data is reused)
            std::vector<int> order[2], chain_off[2];
            chain_off[0].push_back(0);
            chain_off[1].push_back(0);
            // (every block affine -- structured boxes: no chains to form, the affine positions are sorted into CSR order below)
            bool all_affine = want_aff && std::find(cls.begin(), cls.end(), (unsigned char)0) == cls.end();
            if (c->op == FH_MASS_SCALAR && want_aff && !all_affine) {
                // the mass matrix has no kernel for the general positions alone (the generic gather walks every block): a mesh with
                // any non-affine block stays on it entirely
                std::fill(cls.begin(), cls.end(), (unsigned char)0);
                if (cls_d.p) HIP_TRY(c, hipMemsetAsync(cls_d.p, 0, (size_t)nblk, c->stream));
            }
            if (!c->env("FENRIS_HIP_NO_SWEEP") && !all_affine) {
                DevBuf<int> node2blk, succ_d;
                HIP_TRY(c, node2blk.alloc((size_t)N + 1));
                HIP_TRY(c, hipMemsetAsync(node2blk.p, 0xff, sizeof(int) * ((size_t)N + 1), c->stream));  // -1: not in a block
                HIP_TRY(c, succ_d.alloc((size_t)nblk));
                hipLaunchKernelGGL(k_node_to_block, dim3((nblk + 255) / 256), dim3(256), 0, c->stream, c->blk_off.p, nblk, node2blk.p);
                hipLaunchKernelGGL(k_block_successor, dim3(nblk), dim3(64), 0, c->stream, c->gt_hdr.p, c->gt_elems.p, c->conn.p, n,
                                   node2blk.p, nblk, want_aff ? cls_d.p : (const unsigned char*)nullptr, succ_d.p,
                                   c->part_perm ? r2v_d.p : (const int*)nullptr);
                std::vector<int> succ((size_t)nblk);
                HIP_TRY(c, hipMemcpyAsync(succ.data(), succ_d.p, sizeof(int) * (size_t)nblk, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                std::vector<unsigned char> visited((size_t)nblk, 0);
                for (int b = 0; b < nblk; ++b) {
                    if (visited[b]) continue;
                    const int k = cls[b];
                    for (int cur = b; cur >= 0 && cur < nblk && !visited[cur] && cls[cur] == k; cur = succ[cur]) {
                        visited[cur] = 1;
                        order[k].push_back(cur);
                    }
                    chain_off[k].push_back((int)order[k].size());
                }
            } else {
                for (int b = 0; b < nblk; ++b) { order[cls[b]].push_back(b); chain_off[cls[b]].push_back((int)order[cls[b]].size()); }
            }
            mark("successors and chains");
            c->p_cs = us * c->ei.ng;
            c->p_ms = ms;
            c->p_nbs = nb_target;
            c->p_us = us;
            c->p_rw = pipe_record_words(us, ms, n, nb_target);
            // position-indexed tables of one class
            auto build_set = [&](const std::vector<int>& ord, const std::vector<int>& choff, DevBuf<int>& rec, DevBuf<int>& conn,
                                 DevBuf<int>& elem, int by_parity) -> int {
                const int npos = (int)ord.size(), nchains = (int)choff.size() - 1;
                DevBuf<int> order_d, chain_d;
                HIP_TRY(c, order_d.alloc(ord.size()));
                HIP_TRY(c, chain_d.alloc(choff.size()));
                HIP_TRY(c, hipMemcpyAsync(order_d.p, ord.data(), sizeof(int) * ord.size(), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipMemcpyAsync(chain_d.p, choff.data(), sizeof(int) * choff.size(), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, rec.alloc((size_t)npos * c->p_rw));
                HIP_TRY(c, conn.alloc((size_t)npos * c->p_cs));
                HIP_TRY(c, elem.alloc((size_t)npos * us));
#define PT_LAUNCH(NGV)                                                                                                           \
    hipLaunchKernelGGL(k_build_pipe_tables<NGV>, dim3(nchains), dim3(64), 0, c->stream, order_d.p, chain_d.p, c->gt_hdr.p,        \
                       c->gt_elems.p, c->gt_ent.p, c->gt_pos.p, noff_d, c->conn.p, n, c->p_cs, ms, nb_target, us, c->p_rw,     \
                       rec.p, conn.p, elem.p, by_parity)
                switch (c->ei.ng) {
                    case 3: PT_LAUNCH(3); break;
                    case 4: PT_LAUNCH(4); break;
                    case 8: PT_LAUNCH(8); break;
                    default: break;
                }
#undef PT_LAUNCH
                HIP_TRY(c, hipGetLastError());
                HIP_TRY(c, hipStreamSynchronize(c->stream));  // order_d / chain_d are released on return
                return FH_OK;
            };
            c->a_npos = 0;
            if (!order[1].empty()) {
                // the affine kernel keeps nothing staged from one block to the next, and its write-out carries incomplete
                // 128-byte lines from a block to its successor in memory: positions in CSR order, every position its own chain
                std::sort(order[1].begin(), order[1].end());
                chain_off[1].resize(order[1].size() + 1);
                for (size_t k = 0; k <= order[1].size(); ++k) chain_off[1][k] = (int)k;
                DevBuf<int> tmp_rec;  // the pipelined kernel's records: input of the lane builder only
                int rs = build_set(order[1], chain_off[1], tmp_rec, c->a_conn, c->a_elem, 0);
                if (rs) return rs;
                mark("position tables of the affine class (k_build_pipe_tables)");
                const int npos = (int)order[1].size();
                c->a_us = us;
                bool bad = false;
                rs = build_lane_tables(c, tmp_rec.p, us, ms, nb_target, npos, S, c->a_conn.p, c->a_elem.p, c->a_hdr, c->a_lanes, c->a_ntab,
                                       c->a_incomplete, bad, "affine rows");
                if (rs) return rs;
                mark("lane tables of the affine class");
                if (bad) {  // a block the lane tables cannot express: everything on the general kernels
                    c->aff_failed = true;
                    return build_partition(c);
                }
                // (The lane tuner of k_hex8_rows applied to these tables -- element records 80 bytes apart, reference blocks -- was measured:
                // headline 4.91 - 5.06 -> 5.15 - 5.30 ms with the read model alone, +-0 with the staging stores in the model, C2 +4 %.  This
                // kernel is not bound by its LDS reads; the builder's order stays.)
                c->a_conn.release();  // input of the lane builder only
                c->a_npos = npos;
                {   // the element range behind these positions (a node range of a few rows -- the interface plane sent first in a
                    // partition -- needs the records of two element layers, not of ten million elements)
                    DevBuf<int> mm;
                    HIP_TRY(c, mm.alloc(2));
                    const int init[2] = {0x7fffffff, -1};
                    HIP_TRY(c, hipMemcpyAsync(mm.p, init, sizeof init, hipMemcpyHostToDevice, c->stream));
                    const size_t cnt = (size_t)npos * us;
                    hipLaunchKernelGGL(k_minmax_nonneg, dim3((unsigned)std::min<size_t>((cnt + 255) / 256, 4096)), dim3(256), 0, c->stream, c->a_elem.p, cnt, mm.p);
                    int got[2] = {0, -1};
                    HIP_TRY(c, hipMemcpyAsync(got, mm.p, sizeof got, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    c->a_emin = got[1] >= 0 ? got[0] : 0;
                    c->a_emax = got[1];
                }
                mark("element range of the affine class");
                // round 5, the fused form of k_affine_rows: per position the distinct vertices of its slots' elements (nodes 0, 1, 3, 4) and
                // their places per slot, so that the kernel's records wave forms the element records itself (no k_affine_records launch)
                c->a_nu = 0;
                c->a_vtab.release();
                // NOT the default: measured slower than the separate records kernel (profiles/r05_fused_records_experiment.txt); the tables
                // (0.7 GB on the 216^3 mesh) are built only when FENRIS_HIP_AFFINE_FUSED=1 is set before the pattern is built.
                if (c->env_int("FENRIS_HIP_AFFINE_FUSED", 0) != 0 && c->op != FH_MASS_SCALAR) {
                    DevBuf<int> numax;
                    HIP_TRY(c, numax.alloc(1));
                    HIP_TRY(c, hipMemsetAsync(numax.p, 0, sizeof(int), c->stream));
                    HIP_TRY(c, affine_rows_vertex_count(c->stream, c->a_elem.p, c->conn.p, us, npos, numax.p));
                    int nu = 0;
                    HIP_TRY(c, hipMemcpyAsync(&nu, numax.p, sizeof nu, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    const int nu_pad = std::max(4, (nu + 3) / 4 * 4);
                    if (nu > 0 && nu_pad <= 128) {
                        HIP_TRY(c, c->a_vtab.alloc((size_t)npos * (nu_pad + 32)));
                        HIP_TRY(c, affine_rows_vertex_tables(c->stream, c->a_elem.p, c->conn.p, us, npos, c->a_vtab.p, nu_pad));
                        c->a_nu = nu_pad;
                    }
                    mark("vertex tables of the affine class (fused records)");
                }
            }
            c->npos_gen = (int)order[0].size();
            if (!order[0].empty()) {
                int rs = build_set(order[0], chain_off[0], c->p_rec, c->p_conn, c->p_elem, hrows_cand ? 1 : 0);
                if (rs) return rs;
                mark("position tables of the general class");
            }
            c->has_pipe = true;
            if (c->env("FENRIS_HIP_VERBOSE"))
                std::fprintf(stderr, "[fenris_hip] sweep order: %d general blocks in %d chains, %d affine blocks in %d chains (us=%d ms=%d)\n",
                             c->npos_gen, (int)chain_off[0].size() - 1, c->a_npos, (int)chain_off[1].size() - 1, us, ms);
            c->has_rows = false;
            const int npg = c->npos_gen;
            // Hex8, Laplace / uniform LinearElastic: lane tables for the general positions as well (k_hex8_rows, hex8_rows.hip: row-owner
            // lanes instead of LDS atomics; the eight-point rule -- checked at the launch).  Under an element mask a block without an active
            // element gets a lane that stores zeros; when the lanes do not suffice for that somewhere, the pipelined kernel stays.  Its
            // tables are kept: they serve every other rule and per-element parameters.
            c->has_hrows = false;
            if (c->elem_kind == FH_HEX8 && (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC) && us <= HEX8_ROWS_US && nb_target <= 8 && npg > 0 &&
                !c->has_rules && !c->env("FENRIS_HIP_NO_HEX8_ROWS")) {
                bool bad = false;
                int rs = build_lane_tables(c, c->p_rec.p, us, ms, nb_target, npg, S, c->p_conn.p, c->p_elem.p, c->h_hdr, c->h_lanes, c->h_ntab,
                                           c->h_incomplete, bad, "hex8 rows", 1);
                if (rs) return rs;
                if (!bad && !c->h_incomplete) {
                    // lanes rearranged so that the sixteen lanes the LDS serves together read different banks (host, unique tables only)
                    if (c->h_ntab <= c->env_int("FENRIS_HIP_TUNE_LANES_MAX", 4096) && !c->env("FENRIS_HIP_NO_LANE_TUNING")) {
                        std::vector<uint2> tabs((size_t)c->h_ntab * 256);
                        HIP_TRY(c, hipMemcpyAsync(tabs.data(), c->h_lanes.p, sizeof(uint2) * tabs.size(), hipMemcpyDeviceToHost, c->stream));
                        HIP_TRY(c, hipStreamSynchronize(c->stream));
                        double cb = 0.0, ca = 0.0;
                        hex8_rows_tune_lanes(tabs.data(), c->h_ntab, 12345u, &cb, &ca);
                        HIP_TRY(c, hipMemcpyAsync(c->h_lanes.p, tabs.data(), sizeof(uint2) * tabs.size(), hipMemcpyHostToDevice, c->stream));
                        HIP_TRY(c, hipStreamSynchronize(c->stream));
                        if (c->env("FENRIS_HIP_VERBOSE"))
                            std::fprintf(stderr, "[fenris_hip] hex8 rows: %d lane tables tuned, modelled LDS cycles per position and operand sweep %.1f -> %.1f (64 = conflict-free)\n",
                                         c->h_ntab, cb, ca);
                    }
                    HIP_TRY(c, c->h_pos.alloc((size_t)npg * 4));
                    HIP_TRY(c, hex8_rows_positions(c->stream, c->p_rec.p, c->p_rw, us, ms, c->h_hdr.p, npg, c->h_pos.p));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    c->has_hrows = true;
                }
                c->h_hdr.release();   // folded into the position records
                mark("lane tables of the general class (hex8 rows)");
                if (c->env("FENRIS_HIP_VERBOSE"))
                    std::fprintf(stderr, "[fenris_hip] row-owner lanes (Hex8, general positions): %s\n", c->has_hrows ? "built" : "mesh not expressible, pipelined kernel kept");
            }
            // Tet4 with a one-point rule: the row-owner kernel is the default (C3: 1.31 -> 0.85 ms), FENRIS_HIP_NO_ROWS keeps
            // the pipelined kernel
            if (c->elem_kind == FH_TET4 && us * 4 <= 1024 && nb_target <= 16 && npg > 0 && !c->env("FENRIS_HIP_NO_ROWS")) {
                c->r_rw = 8 + us / 4 + nb_target + 1 + nb_target;
                DevBuf<unsigned> row_real;   // first entry of every node's real row, in the order of the blocks
                HIP_TRY(c, row_real.alloc((size_t)N + 1));
                hipLaunchKernelGGL(k_row_starts, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->noff.p,
                                   c->part_perm ? v2r_d.p : (const unsigned*)nullptr, N, row_real.p);
                DevBuf<int> st;
                HIP_TRY(c, st.alloc(1));
                HIP_TRY(c, hipMemsetAsync(st.p, 0, sizeof(int), c->stream));
                HIP_TRY(c, c->r_rec.alloc((size_t)npg * c->r_rw));
                int bad = 0;
                for (int ls : {128, 256}) {  // half the table (and its traffic) when no block needs more than 128 lanes
                    c->r_ls = ls;
                    HIP_TRY(c, hipMemsetAsync(st.p, 0, sizeof(int), c->stream));
                    HIP_TRY(c, c->r_lanes4.alloc((size_t)npg * ls));
                    hipLaunchKernelGGL(k_build_row_lanes_tet4, dim3(npg), dim3(64), 0, c->stream, c->p_rec.p, c->p_rw, us, ms,
                                       nb_target, npg, c->r_rw, c->r_rec.p, c->r_lanes4.p, ls, st.p, row_real.p);
                    HIP_TRY(c, hipGetLastError());
                    HIP_TRY(c, hipMemcpyAsync(&bad, st.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    if (bad != 2) break;  // 2: only the stride was too small
                }
                mark("row lanes (Tet4)");
                if (bad == 0) {   // the position's unique vertices and the slot words that index them
                    HIP_TRY(c, hipMemsetAsync(st.p, 0, sizeof(int), c->stream));
                    HIP_TRY(c, c->r_vconn.alloc((size_t)npg * (ROWS_TET4_VMAX + us)));
                    hipLaunchKernelGGL(k_build_row_verts_tet4, dim3(npg), dim3(64), 0, c->stream, c->p_conn.p, us, npg, c->r_vconn.p, st.p);
                    HIP_TRY(c, hipGetLastError());
                    HIP_TRY(c, hipMemcpyAsync(&bad, st.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                }
                c->has_rows = bad == 0;
                HIP_TRY(c, hipStreamSynchronize(c->stream));  // row_real is released at the end of this scope
                mark("row vertices (Tet4)");
                if (c->env("FENRIS_HIP_VERBOSE"))
                    std::fprintf(stderr, "[fenris_hip] row-owner lanes (Tet4, stride %d): %s\n", c->r_ls,
                                 c->has_rows ? "built" : "mesh not expressible, pipelined kernel kept");
            }
        }
    }
    if (c->part_rows_only && !c->has_rows) {  // these tables serve the row-owner kernel only: smaller blocks, then the standard form
        if (++c->rows_try >= 2) c->perm_failed = true;
        c->part_perm = false;
        c->part_rows_only = false;
        return build_partition(c);
    }
    c->has_slotpar = false;
    if (c->has_rules && c->fast_ok && c->op != FH_LAPLACE && !c->has_pipe) {
        // per-element data without the pipelined tables: the generic kernels need the per-point-coefficient layout
        c->fast_ok = false;
        c->elem_par = false;
        return build_partition(c);
    }
    c->g_ub = ub;
    c->g_umax = umax;
    c->g_mb = mb;
    c->g_acc = acc;
    c->g_nb = 64;
    c->has_partition = true;
    mark("the rest");
    return FH_OK;
}

template <int OP, bool ELEMPAR = false>
int launch_rows_tet4(fh_ctx* c, KArgs& a, const RowTablesS& T) {
    // the layout's integers + two parities of the record + the slot words
    const size_t lds = make_layout<FH_TET4, OP, WHAT_MATRIX>(a.nq, a.ub, 0, a.nb_max, true, 0, 1, 1, 0, 2).bytes() +
                       sizeof(int) * (size_t)(2 * T.rw + T.us + 4);
    if (lds > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "row-owner gather: LDS footprint too large");
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
    // workgroups per CU, measured inside one context on the same buffers (scripts/ab_in_context.py, C3): elasticity 2 (0.562 ms; 3: 0.594,
    // 4: 0.603, 5: 0.585), Laplace 4
    const size_t cap = (c->op == FH_LAPLACE) ? 4 : 2;
    const int per_cu = std::max(1, (int)std::min<size_t>(cap, (LDS_LIMIT - 512) / std::max<size_t>(lds, 1)));
    // (FENRIS_HIP_PIPE_GRID: tests force many positions per workgroup on small meshes)
    const int grid = std::max(1, std::min(c->npos_gen, c->env_int("FENRIS_HIP_PIPE_GRID", dev_cus * c->env_int("FENRIS_HIP_PIPE_WGS_PER_CU", per_cu))));
    auto kern = a.trace ? k_gather_rows_tet4<OP, ELEMPAR, true> : k_gather_rows_tet4<OP, ELEMPAR>;   // FENRIS_HIP_TRACE: instrumented twin
    if (lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (c->env("FENRIS_HIP_VERBOSE"))
        std::fprintf(stderr, "[fenris_hip] row-owner gather (Tet4): lds=%zu B wgs/cu=%d grid=%d\n", lds, per_cu, grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, c->stream, a, T);
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

template <int EK, int OP, int QC, int JT>
int launch_pipelined_j(fh_ctx* c, KArgs& a, const PipeTables& T) {
    size_t lds = make_layout<EK, OP, WHAT_MATRIX>(a.nq, a.ub, a.acc_max, a.nb_max, true, a.mb, 1, QC).bytes();
    if (lds > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "pipelined gather: LDS footprint too large");
    // the compile-time-rule instantiation (Hex8, rule of exactly QC points) stages planar gradient rows, which are
    // longer: taken only while two workgroups still share a CU
    bool fullq = false;
    if constexpr (EK == FH_HEX8 && QC == 8 && JT == 2) {
        const size_t lds_planar = make_layout<EK, OP, WHAT_MATRIX>(a.nq, a.ub, a.acc_max, a.nb_max, true, a.mb, 1, QC, 0, 1).bytes();
        fullq = a.nq == QC && T.cs <= 256 && T.rw <= 256 && !c->env("FENRIS_HIP_NO_FULLQ") &&
                (2 * lds_planar + 1024 <= LDS_LIMIT || 2 * lds + 1024 > LDS_LIMIT);
        if (fullq) lds = lds_planar;
    }
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
    const int per_cu = std::max(1, (int)std::min<size_t>(8, (LDS_LIMIT - 512) / std::max<size_t>(lds, 1)));
    const int wgs = std::max(1, c->env_int("FENRIS_HIP_PIPE_WGS_PER_CU", per_cu));
    const int grid = std::max(1, std::min(c->npos_gen, c->env_int("FENRIS_HIP_PIPE_GRID", dev_cus * wgs)));
    // the instrumented instantiation only where it is used for profiling (Hex8, the default tiling)
    const bool dbg = (c->env("FENRIS_HIP_TRACE") || c->env("FENRIS_HIP_ABLATE") || c->env("FENRIS_HIP_DBG_KERNEL"));
    void (*kern)(const KArgs, const PipeTables) = k_gather_pipelined<EK, OP, QC, JT>;
    constexpr int N_ = ElemT<EK>::N;
    constexpr bool DEFAULT_JT = JT == ((N_ % 2 == 0) ? 2 : N_);  // per-element data: the default tiling only
    if (T.slotpar) {
        if constexpr (OP == FH_LINEAR_ELASTIC && DEFAULT_JT) {
            kern = k_gather_pipelined<EK, OP, QC, JT, false, false, true>;
            if constexpr (EK == FH_HEX8 && QC == 8 && JT == 2)
                if (fullq) kern = k_gather_pipelined<EK, OP, QC, JT, false, true, true>;
        } else {
            return c->fail(FH_UNSUPPORTED, "pipelined gather with per-element parameters: default FENRIS_HIP_PIPE_JT only");
        }
    } else if constexpr (EK == FH_TET4 && QC == 1 && JT == 2) {
        if (dbg) kern = k_gather_pipelined<EK, OP, QC, JT, true>;
    } else if constexpr (EK == FH_HEX8 && QC == 8 && JT == 2) {
        if (dbg) {  // the instrumented twin of whichever instantiation production would take
            if (fullq) kern = k_gather_pipelined<EK, OP, QC, JT, true, true>;
            else kern = k_gather_pipelined<EK, OP, QC, JT, true>;
        } else if (fullq)
            kern = k_gather_pipelined<EK, OP, QC, JT, false, true>;
    }
    if (lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (c->env("FENRIS_HIP_VERBOSE"))
        std::fprintf(stderr, "[fenris_hip] pipelined gather: QC=%d JT=%d lds=%zu B wgs/cu=%d grid=%d\n", QC, JT, lds, wgs, grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, c->stream, a, T);
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

template <int EK, int OP, int QC>
int launch_pipelined_q(fh_ctx* c, KArgs& a, const PipeTables& T) {
    constexpr int N = ElemT<EK>::N;
    const int jt = c->p_jt;
    if (N % 4 == 0 && jt == 4) return launch_pipelined_j<EK, OP, QC, 4>(c, a, T);
    if (N % 2 == 0 && jt == 2) return launch_pipelined_j<EK, OP, QC, 2>(c, a, T);
    if (jt == N) return launch_pipelined_j<EK, OP, QC, N>(c, a, T);
    return launch_pipelined_j<EK, OP, QC, 1>(c, a, T);
}

template <int EK, int OP>
int launch_pipelined_t(fh_ctx* c, KArgs& a, const PipeTables& T, size_t, int) {
    // staged quadrature points per chunk: the largest chunk (not larger than the rule) that still lets >= 2
    // workgroups share a CU (measured on Hex8: profiles/r01_sweep_128_pipelined_nb_qc_jt.txt)
    int qc = c->env_int("FENRIS_HIP_PIPE_QC", 0);
    if (qc <= 0) {
        qc = 1;
        for (int cand : {8, 4, 2}) {
            if (cand > a.nq && cand > 1 && cand / 2 >= a.nq) continue;  // would stage empty slots
            const size_t lds = make_layout<EK, OP, WHAT_MATRIX>(a.nq, a.ub, a.acc_max, a.nb_max, true, a.mb, 1, cand).bytes();
            if (2 * lds + 1024 <= LDS_LIMIT) { qc = cand; break; }
        }
    }
    if (a.nq == 1) qc = 1;
    if (qc >= 8) return launch_pipelined_q<EK, OP, 8>(c, a, T);
    if (qc >= 4) return launch_pipelined_q<EK, OP, 4>(c, a, T);
    if (qc >= 2) return launch_pipelined_q<EK, OP, 2>(c, a, T);
    return launch_pipelined_q<EK, OP, 1>(c, a, T);
}

int launch_pipelined(fh_ctx* c, KArgs& a, const PipeTables& T, size_t lds, int grid) {
    const bool lap = c->op == FH_LAPLACE;
    switch (c->elem_kind) {
        case FH_HEX8: return lap ? launch_pipelined_t<FH_HEX8, FH_LAPLACE>(c, a, T, lds, grid)
                                 : launch_pipelined_t<FH_HEX8, FH_LINEAR_ELASTIC>(c, a, T, lds, grid);
        case FH_TET4: return lap ? launch_pipelined_t<FH_TET4, FH_LAPLACE>(c, a, T, lds, grid)
                                 : launch_pipelined_t<FH_TET4, FH_LINEAR_ELASTIC>(c, a, T, lds, grid);
        case FH_QUAD4: return lap ? launch_pipelined_t<FH_QUAD4, FH_LAPLACE>(c, a, T, lds, grid)
                                  : launch_pipelined_t<FH_QUAD4, FH_LINEAR_ELASTIC>(c, a, T, lds, grid);
        case FH_TRI3: return lap ? launch_pipelined_t<FH_TRI3, FH_LAPLACE>(c, a, T, lds, grid)
                                 : launch_pipelined_t<FH_TRI3, FH_LINEAR_ELASTIC>(c, a, T, lds, grid);
        default: return c->fail(FH_UNSUPPORTED, "pipelined gather: unsupported element");
    }
}

// node blocks all of whose elements are affine: k_affine_ring / k_affine_rows (affine_ring.hip, affine_rows.hip) over their position tables
int launch_affine(fh_ctx* c, KArgs& a) {
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
    // the scalar mass matrix rides the Laplace kernel: records (|det J|, 0 ...), reference blocks (sum_q w rho phi_a phi_b, 0 ...)
    const int rop = (c->op == FH_MASS_SCALAR) ? (int)FH_LAPLACE : c->op;
    const int gw = (rop == FH_LAPLACE) ? AFFINE_ROWS_GW_LAP : AFFINE_ROWS_GW_LE;
    // round 5 (experiment, FENRIS_HIP_AFFINE_FUSED=1): the element records formed inside k_affine_rows by a seventh wavefront (FUSED
    // instantiation) -- no k_affine_records launch, no record array.  Measured SLOWER than the two launches (4.82 against 4.73 ms on the
    // headline in one context, C2 0.56 against 0.31): what the records kernel costs is its cold reads, and the fused form has as many.
    const int a_depth = c->env_int("FENRIS_HIP_AFFINE_DEPTH", 2), a_nstore = c->env_int("FENRIS_HIP_AFFINE_STORE_WAVES", 1);
    const int a_chunk = c->env_int("FENRIS_HIP_AFFINE_CHUNK", 0);
    bool fused = c->op != FH_MASS_SCALAR && c->a_nu > 0 && c->a_vtab.p && c->env_int("FENRIS_HIP_AFFINE_FUSED", 0) != 0 &&
                 affine_rows_can_fuse(a_depth, a_nstore, a.ablate, a_chunk);
    if (fused && affine_rows_lds_bytes(rop, c->a_us, c->g_acc, c->a_nu) > LDS_LIMIT) fused = false;
#ifdef FENRIS_HIP_WITH_RING
    if (c->env_int("FENRIS_HIP_AFFINE_RING", 0) != 0) fused = false;
#endif
    if (!fused && c->a_recs.n < (size_t)c->E * gw) HIP_TRY(c, c->a_recs.alloc((size_t)c->E * gw));
    const unsigned char* act = c->has_mask ? c->active.p : nullptr;
    DevStatus* status = c->status.p + c->status_slot;
    const int nt = (c->env_int("FENRIS_HIP_AFFINE_NT", rop == FH_LAPLACE ? 1 : 0) ? AFFINE_ROWS_NT_STORES : 0) |
                   (c->env("FENRIS_HIP_AFFINE_NO_CARRY") ? AFFINE_ROWS_NO_CARRY : 0) | (c->env("FENRIS_HIP_AFFINE_NO_CLEAR") ? AFFINE_ROWS_NO_CLEAR : 0) |
                   ((c->env_int("FENRIS_HIP_AFFINE_REC_ABLATE", 0) & 1) ? AFFINE_ROWS_REC_NO_DMA : 0) | ((c->env_int("FENRIS_HIP_AFFINE_REC_ABLATE", 0) & 2) ? AFFINE_ROWS_REC_NO_MATH : 0) |
                   ((c->env_int("FENRIS_HIP_AFFINE_REC_ABLATE", 0) & 4) ? AFFINE_ROWS_REC_NO_L1 : 0) | ((c->env_int("FENRIS_HIP_AFFINE_REC_ABLATE", 0) & 8) ? AFFINE_ROWS_REC_NO_L2 : 0);
    // third form (affine_ring.hip): no barrier in the sweep, rows staged in a ring; second form: one barrier per position, double buffer
    // (instrumentation: compiled only into a `make TRACE=1` library)
#ifdef FENRIS_HIP_WITH_RING
    const bool use_ring = c->env_int("FENRIS_HIP_AFFINE_RING", 0) != 0;
#endif
    if (c->env("FENRIS_HIP_VERBOSE_PTRS"))   // where the buffers of this context lie (the spread between identical contexts, profiles/r03_affine_experiments.txt)
        std::fprintf(stderr, "[fenris_hip ptrs] recs=%p hdr=%p elem=%p lanes=%p vals=%p verts=%p conn=%p\n", (void*)c->a_recs.p, (void*)c->a_hdr.p,
                     (void*)c->a_elem.p, (void*)c->a_lanes.p, (void*)a.vals, (void*)c->verts.p, (void*)c->conn.p);
    auto rows = [&](int pos0, int count) -> int {
        AffineRowTables T{c->a_hdr.p, c->a_lanes.p, c->a_elem.p, c->a_recs.p,
                          c->ghat.p + (c->op == FH_MASS_SCALAR ? 64 * (AFFINE_GW_LE + AFFINE_GW_LAP) : c->op == FH_LAPLACE ? 64 * AFFINE_GW_LE : 0), c->a_us, count,
                          c->g_acc, pos0, c->a_npos, c->a_incomplete, a_chunk, fused ? c->a_vtab.p : nullptr, fused ? c->a_nu : 0};
#ifdef FENRIS_HIP_WITH_RING
        if (use_ring) {
            const int ring = affine_ring_doubles(c->g_acc, c->env_int("FENRIS_HIP_AFFINE_RING_KB", 0));
            const size_t lds = affine_ring_lds_bytes(rop, c->a_us, ring);
            if (lds <= LDS_LIMIT) {
                const int cap = rop == FH_LAPLACE ? 4 : 3;
                const int per_cu = std::max(1, (int)std::min<size_t>(cap, LDS_LIMIT / std::max<size_t>(lds, 1)));
                const int grid = std::max(1, std::min(count, c->env_int("FENRIS_HIP_AFFINE_GRID", dev_cus * c->env_int("FENRIS_HIP_AFFINE_WGS_PER_CU", per_cu))));
                if (c->env("FENRIS_HIP_VERBOSE"))
                    std::fprintf(stderr, "[fenris_hip] affine ring: positions %d + %d ring=%d doubles lds=%zu B wgs/cu=%d grid=%d\n", pos0, count, ring, lds, per_cu, grid);
                HIP_TRY(c, affine_ring_launch(rop, ring, c->env_int("FENRIS_HIP_AFFINE_DEPTH", 2), grid, lds, c->stream, a, T,
                                              a.ablate | nt | ((c->env_int("FENRIS_HIP_AFFINE_THROTTLE", 0) & 0xff) << 20) | ((c->env_int("FENRIS_HIP_AFFINE_PRIO", 0) & 3) << 28)));
                return FH_OK;
            }
        }
#endif
        const size_t lds = affine_rows_lds_bytes(rop, c->a_us, c->g_acc, fused ? c->a_nu : 0);
        if (lds > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "affine gather: LDS footprint too large");
        // workgroups per CU, measured best: 3 (elasticity), 4 (Laplace: fewer registers, less LDS)
        const int per_cu = std::max(1, (int)std::min<size_t>(rop == FH_LAPLACE ? 4 : 3, (LDS_LIMIT - 512) / std::max<size_t>(lds, 1)));
        const int grid = std::max(1, std::min(count, c->env_int("FENRIS_HIP_AFFINE_GRID", dev_cus * c->env_int("FENRIS_HIP_AFFINE_WGS_PER_CU", per_cu))));
        if (c->env("FENRIS_HIP_VERBOSE"))
            std::fprintf(stderr, "[fenris_hip] affine rows: positions %d + %d lds=%zu B wgs/cu=%d grid=%d fused=%d\n", pos0, count, lds, per_cu, grid, (int)fused);
        HIP_TRY(c, affine_rows_launch(rop, a_depth, a_nstore, grid, lds, c->stream, a, T, a.ablate | nt, c->has_mask, fused));
        return FH_OK;
    };
    if (fused) return rows(0, c->a_npos);
    // element records first (R = sqrt|det J| J^-1 or M = R R^T per affine element): same stream, once per assembly.  (Round 3: making
    // the records of all but the first eighth of the sweep on a second stream beside the first part's launch was measured 0.3 ms
    // SLOWER than the 0.41 ms it hides -- the two kernels' workgroups compete for the CUs; two launches of the sweep in one stream cost
    // nothing measurable, and records made chunk by chunk right before their part of the sweep (to be read back from the memory-side
    // cache) change nothing up to 4 chunks and lose from 8 on.  profiles/r03_affine_experiments.txt)
    HIP_TRY(c, affine_records_launch(c->op, c->stream, c->verts.p, c->conn.p, c->elem_aff.p, act, c->a_emin, std::min<long long>(c->a_emax + 1, (long long)c->E),
                                     c->a_recs.p, status));
    return rows(0, c->a_npos);
}

// dense element matrices of the elements [first, first + count) into device memory (no status read-back)
static int element_matrices_enqueue(fh_ctx* c, uint64_t first, uint64_t count, double* ke_dev, bool by_elem) {
    KArgs a;
    fill_common(c, a);
    a.ke_out = ke_dev;
    a.ke_by_elem = by_elem ? 1 : 0;
    a.labels = (by_elem && c->has_mask) ? c->active_list.p : nullptr;  // two-pass assembly: the active elements only
    a.work_begin = (long long)first;
    a.work_end = (long long)(first + count);
    a.epb = choose_epb(c, WHAT_MATRIX);
    a.ub = a.epb;
    const size_t lds = layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, a.ub, 0, 0, false, 0, a.fast);
    if (lds > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "quadrature rule too large for LDS staging");
    const int grid = (int)((count + a.epb - 1) / a.epb);
    int rc = FH_OK;
#define CALL(EKC, OPC) rc = launch_matrix<EKC, OPC>(c, a, MODE_DUMP, lds, grid)
    FH_FOR_ELEM_OP(c->elem_kind, c->op, CALL)
#undef CALL
    return rc;
}

// Owner-computes for high-order elements (n > 8), two passes: dense element matrices (element-parallel, every K_e
// computed once), then one wavefront per node gathers the columns of its elements' K_e into its CSR rows.
// Recomputing K_e per owning node block, as the one-pass kernels do, costs 8-27x for a 27-node element.
int assemble_two_pass(fh_ctx* c, double* values_dev, int overwrite) {
    const int S = c->S();
    const size_t ld = (size_t)S * c->ei.n;
    if (c->ke_dense.n < ld * ld * c->E) HIP_TRY(c, c->ke_dense.alloc(ld * ld * c->E));
    // first pass: Hex27 LinearElastic / NeoHookean with a uniform table run on the matrix cores (hex27_mfma.hpp) and
    // write the planar layout; everything else takes the generic element kernel (column-major K_e)
    const bool mfma = c->elem_kind == FH_HEX27 && (c->op == FH_LINEAR_ELASTIC || c->op == FH_NEO_HOOKEAN) && !c->has_rules &&
                      c->nq == 27 && c->has_params && !c->env("FENRIS_HIP_NO_MFMA");
    int rc = FH_OK;
    if (mfma) {
        KArgs a;
        fill_common(c, a);
        a.ke_out = c->ke_dense.p;
        a.labels = c->has_mask ? c->active_list.p : nullptr;
        a.work_begin = 0;
        a.work_end = (long long)(c->has_mask ? c->num_active : c->E);
        const size_t lds1 = sizeof(double) * (size_t)Hex27Lds::total;
        int dev_cus = 256;
        (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
        // (FENRIS_HIP_TWO_PASS_GRID: tests force many elements / nodes per workgroup on small meshes)
        const int grid1 = std::max(1, (int)std::min<long long>(a.work_end, c->env_int("FENRIS_HIP_TWO_PASS_GRID", dev_cus * std::max(1, c->env_int("FENRIS_HIP_HEX27_WGS_PER_CU", 2)))));
        if (grid1 > 0) {
            if (c->op == FH_NEO_HOOKEAN && a.trace) {   // FENRIS_HIP_TRACE: per-phase cycle counters
                auto kern = k_hex27_dense_mfma<FH_NEO_HOOKEAN, true>;
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
                hipLaunchKernelGGL(kern, dim3(grid1), dim3(256), lds1, c->stream, a, c->uni_mu, c->uni_lambda);
            } else if (c->op == FH_NEO_HOOKEAN) {
                auto kern = k_hex27_dense_mfma<FH_NEO_HOOKEAN>;
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
                hipLaunchKernelGGL(kern, dim3(grid1), dim3(256), lds1, c->stream, a, c->uni_mu, c->uni_lambda);
            } else {
                auto kern = k_hex27_dense_mfma<FH_LINEAR_ELASTIC>;
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
                hipLaunchKernelGGL(kern, dim3(grid1), dim3(256), lds1, c->stream, a, c->uni_mu, c->uni_lambda);
            }
            HIP_TRY(c, hipGetLastError());
        }
    } else {
        rc = element_matrices_enqueue(c, 0, c->has_mask ? c->num_active : c->E, c->ke_dense.p, true);
        if (rc) return rc;
    }
    const unsigned max_row = c->max_row;  // longest node row, cached with the pattern (no O(N) host scan per assembly)
    const size_t lds = (size_t)4 * sizeof(double) * S * S * max_row;
    if (lds > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "two-pass gather: a node row does not fit in LDS");
    if (max_row >= 65536) return c->fail(FH_UNSUPPORTED, "two-pass gather: node valence too large");
    const unsigned* adj_off = c->has_mask ? c->n2e_off_c.p : c->n2e_off.p;
    const unsigned* adj = c->has_mask ? c->n2e_c.p : c->n2e.p;
    { const int rc_h = host_offsets(c); if (rc_h) return rc_h; }
    const std::vector<unsigned>& adj_off_h = c->has_mask ? c->h_n2e_off_c : c->h_n2e_off;
    const long long entries = adj_off_h.empty() ? 0 : (long long)adj_off_h[c->N];
    const bool wide = max_row >= 256;
    if (!c->has_tp_pos) {  // once per pattern / element mask
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
    }
    const int grid = std::max(1, (int)std::min<uint64_t>((c->N + 3) / 4, (uint64_t)c->env_int("FENRIS_HIP_TWO_PASS_ROWS_GRID", c->env_int("FENRIS_HIP_TWO_PASS_GRID", 1 << 17))));   // (C4: 2^17 workgroups 8.33 ms, one per four nodes (410 k) 8.42, 2^13 8.45, 2^11 8.68)
    c->last_kernel = mfma ? "k_hex27_dense_mfma + k_rows_from_dense" : "k_assemble_matrix<dump> + k_rows_from_dense";
#define ROWS(SS, PT, PTR)                                                                                                     \
    do {                                                                                                                       \
        if (mfma) { ROWS2(SS, PT, PTR, true); } else { ROWS2(SS, PT, PTR, false); }                                            \
    } while (0)
#define ROWS2(SS, PT, PTR, PL)                                                                                                \
    do {                                                                                                                       \
        void (*kern)(int, int, const unsigned*, const unsigned*, const unsigned*, const PT*, const double*, double*, int, int) = \
            k_rows_from_dense<SS, PT, PL>;                                                                                     \
        if (!PL && !c->env("FENRIS_HIP_NO_ROWS_SMALL")) {                                                                 \
            const int ld_ = SS * (int)c->ei.n;                                                                                 \
            if (ld_ <= 8) kern = k_rows_from_dense_small<SS, PT, 8>;                                                           \
            else if (ld_ <= 16) kern = k_rows_from_dense_small<SS, PT, 16>;                                                    \
            else if (ld_ <= 32) kern = k_rows_from_dense_small<SS, PT, 32>;                                                    \
        }                                                                                                                      \
        if (lds > 48 * 1024)                                                                                                   \
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, c->stream, (int)c->N, c->ei.n, c->noff.p, adj_off, adj, PTR,         \
                           c->ke_dense.p, values_dev, overwrite, (int)max_row);                                                \
    } while (0)
    if (wide) { if (S == 1) ROWS(1, unsigned short, c->tp_pos16.p); else if (S == 2) ROWS(2, unsigned short, c->tp_pos16.p); else ROWS(3, unsigned short, c->tp_pos16.p); }
    else      { if (S == 1) ROWS(1, unsigned char, c->tp_pos8.p); else if (S == 2) ROWS(2, unsigned char, c->tp_pos8.p); else ROWS(3, unsigned char, c->tp_pos8.p); }
#undef ROWS
#undef ROWS2
    HIP_TRY(c, hipGetLastError());
    return FH_OK;
}

int assemble_matrix_enqueue(fh_ctx* c, double* values_dev, int flags, bool reset = true) {
    int rc = check_ready(c, "fh_assemble_matrix", true);
    if (rc) return rc;
    if (!values_dev) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_matrix: values is null");
    const int mode = flags & FH_SCATTER_MASK;
    const int overwrite = (flags & FH_ASSEMBLE_OVERWRITE) ? 1 : 0;
    if (reset) rc = reset_status(c);
    if (rc) return rc;
    if (c->E == 0) return FH_OK;
    KArgs a;
    fill_common(c, a);
    a.vals = values_dev;
    a.overwrite = overwrite;
    const uint64_t nnz = (uint64_t)c->S() * c->S() * c->nnz_nodes;
    if (mode == FH_SCATTER_GATHER && c->row_hi < 0 && !c->env("FENRIS_HIP_NO_TWO_PASS")) {
        // two-pass owner-computes (dense element matrices, then a row gather) where recomputing the element prologue per
        // owning node block is the expensive part: high-order elements, and the nonlinear materials on any element
        // (measured, Hex8 128^3: NeoHookean 11.1 -> 9.2 ms, StVK 19.2 -> 10.0 ms; LinearElastic with per-point
        // parameters is faster one-pass: 5.7 vs 8.2 ms).  The dense buffer costs E (s n)^2 doubles: capped.
        const size_t ld = (size_t)c->S() * c->ei.n;
        const double dense_gb = (double)c->E * ld * ld * 8.0 / 1e9;
        const bool want = c->ei.n > 8 || c->op == FH_NEO_HOOKEAN || c->op == FH_STVK || c->env("FENRIS_HIP_TWO_PASS");
        if (want && dense_gb <= (double)c->env_int("FENRIS_HIP_TWO_PASS_MAX_GB", 96)) {
            // the dense buffer is allocated here: when the device cannot hold it the one-pass gather below takes over
            if (c->ke_dense.n >= ld * ld * c->E || c->ke_dense.alloc(ld * ld * c->E) == hipSuccess)
                return assemble_two_pass(c, values_dev, overwrite);
            (void)hipGetLastError();
        }
    }
    if (mode == FH_SCATTER_GATHER) {
        rc = build_partition(c);
        if (rc) return rc;
        if (c->part_rows_only && !(c->has_pipe && c->has_rows && a.fast && (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC) &&
                              !c->env("FENRIS_HIP_TRACE"))) {
            // these tables are for the row-owner kernel only (see build_partition); another kernel is about to run
            c->perm_failed = true;
            c->has_partition = false; ++c->struct_gen;
            rc = build_partition(c);
            if (rc) return rc;
        }
        if (c->nblk == 0) return FH_OK;  // empty row range
        c->last_kernel.clear();
        if (c->a_npos > 0) {
            // node blocks whose elements are all affine (affine_kernel.hpp); the remaining positions follow below
            rc = launch_affine(c, a);
            if (rc) return rc;
            c->last_kernel = "k_affine_rows";
            if (c->npos_gen == 0) return FH_OK;
            c->last_kernel += " + ";
        }
        a.blk_off = c->blk_off.p;
        a.gt_hdr = c->gt_hdr.p;
        a.gt_elems = c->gt_elems.p;
        a.gt_ent = c->gt_ent.p;
        a.gt_pos = c->has_pos ? c->gt_pos.p : nullptr;
        a.nblk = c->nblk;
        a.ub = c->g_ub;
        a.mb = c->g_mb;
        a.acc_max = c->g_acc;
        a.nb_max = c->g_nb;
        const size_t lds = layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, a.ub, a.acc_max, a.nb_max, true, a.mb, a.fast);
        const bool pipe_rules = c->has_pipe && c->has_rules && c->elem_par && c->fast_ok && c->op == FH_LINEAR_ELASTIC;
        if (pipe_rules && !c->has_slotpar) {
            const size_t n = (size_t)c->npos_gen * c->p_us;
            HIP_TRY(c, c->p_slotpar.alloc(2 * n));
            hipLaunchKernelGGL(k_build_slot_params, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->p_elem.p, n,
                               c->rule_map.p, c->rparams.p, c->nq, c->p_slotpar.p);
            HIP_TRY(c, hipGetLastError());
            c->has_slotpar = true;
        }
        // Tet4 is affine: gradients and det J are the same at every point, so with uniform parameters any rule equals the
        // one-point rule that carries the sum of its weights (the table of gradients at point 0 serves as is)
        if (c->has_pipe && c->has_rows && c->elem_kind == FH_TET4 && (a.fast || pipe_rules) &&
            (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC)) {
            a.fast = 1;
            if (c->nq > 1) {
                a.qw = c->qw.p + c->nq;
                a.nq = 1;
            }
            RowTablesS T{c->r_rec.p, c->r_lanes4.p, c->r_vconn.p, c->p_elem.p, pipe_rules ? c->p_slotpar.p : nullptr,
                         c->r_rw, c->p_us, c->p_nbs, c->npos_gen, c->r_ls};
            a.ub = std::max(c->p_us, 76);   // the X region of the layout (14 doubles per slot) holds the vertex table: 256 x 4 doubles
            a.nb_max = c->p_nbs;
            if (c->has_mask && a.overwrite) {   // blocks without an active element have no lane: clear the range first (rows_kernel.hpp)
                const int n_lo = (c->row_hi < 0) ? 0 : (int)std::min<long long>(c->row_lo, (long long)c->N);
                const int n_hi = (c->row_hi < 0) ? (int)c->N : (int)std::min<long long>(c->row_hi, (long long)c->N);
                if (n_hi > n_lo) {
                    hipLaunchKernelGGL(k_zero_node_rows, dim3(2048), dim3(256), 0, c->stream, c->noff.p, n_lo, n_hi, c->S() * c->S(), values_dev);
                    HIP_TRY(c, hipGetLastError());
                }
            }
            c->last_kernel += "k_gather_rows";
            if (pipe_rules) return launch_rows_tet4<FH_LINEAR_ELASTIC, true>(c, a, T);
            return c->op == FH_LAPLACE ? launch_rows_tet4<FH_LAPLACE>(c, a, T) : launch_rows_tet4<FH_LINEAR_ELASTIC>(c, a, T);
        }
        if (c->has_pipe && c->has_hrows && a.fast && !pipe_rules && c->nq == 8 && c->elem_kind == FH_HEX8 &&
            (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC) && !c->env("FENRIS_HIP_NO_HEX8_ROWS")) {
            const size_t lds_h = hex8_rows_lds_bytes(c->g_acc);
            if (lds_h <= LDS_LIMIT) {
                int dev_cus = 256;
                (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
                const int per_cu = std::max(1, (int)std::min<size_t>(2, (LDS_LIMIT - 512) / std::max<size_t>(lds_h, 1)));
                const int grid = std::max(1, std::min(c->npos_gen, c->env_int("FENRIS_HIP_PIPE_GRID", dev_cus * c->env_int("FENRIS_HIP_PIPE_WGS_PER_CU", per_cu))));
                Hex8RowTables T{c->h_pos.p, c->h_lanes.p, c->p_conn.p, c->p_elem.p, c->p_us, c->p_cs, c->npos_gen, c->g_acc};
                if (c->env("FENRIS_HIP_VERBOSE"))
                    std::fprintf(stderr, "[fenris_hip] hex8 rows: positions %d lds=%zu B wgs/cu=%d grid=%d\n", c->npos_gen, lds_h, per_cu, grid);
                c->last_kernel += "k_hex8_rows";
                HIP_TRY(c, hex8_rows_launch(c->op, grid, lds_h, c->stream, a, T, a.ablate | (a.trace ? 0x10000 : 0)));
                return FH_OK;
            }
        }
        if (c->has_pipe && (a.fast || pipe_rules) && (c->op == FH_LAPLACE || c->op == FH_LINEAR_ELASTIC)) {
            a.fast = 1;
            PipeTables T{c->p_rec.p, c->p_conn.p, c->p_elem.p, pipe_rules ? c->p_slotpar.p : nullptr, c->p_rw,
                         c->p_cs, c->p_ms, c->p_nbs, c->p_us, c->npos_gen};
            a.ub = c->p_us;  // LDS slots: every unique element of a block is staged, shared ones persist
            a.mb = c->p_ms;  // the LDS layout is sized by the table strides
            a.nb_max = c->p_nbs;
            c->last_kernel += "k_gather_pipelined";
            return launch_pipelined(c, a, T, 0, 0);
        }
        c->last_kernel += "k_assemble_matrix<gather>";
#define CALL(EKC, OPC) rc = launch_matrix<EKC, OPC>(c, a, MODE_GATHER, lds, c->nblk)
        FH_FOR_ELEM_OP(c->elem_kind, c->op, CALL)
#undef CALL
        return rc;
    }
    if (c->row_hi >= 0) return c->fail(FH_UNSUPPORTED, "fh_assemble_matrix: a row range needs FH_SCATTER_GATHER");
    if (overwrite) HIP_TRY(c, hipMemsetAsync(values_dev, 0, sizeof(double) * nnz, c->stream));
    a.epb = choose_epb(c, WHAT_MATRIX);
    a.ub = a.epb;
    // high-order elements: column search of the scatter in LDS (neighbour lists staged per element)
    if (c->ei.n > 8 && !c->env("FENRIS_HIP_NO_NC_LDS")) {
        const unsigned max_row = c->max_row;
        const size_t with_nc = layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, a.ub, 0, 0, false, 0, a.fast, (int)max_row);
        if (with_nc <= LDS_TARGET + 16 * 1024) a.nc_row = (int)max_row;
    }
    const size_t lds = layout_bytes_dyn(c->elem_kind, c->op, WHAT_MATRIX, c->nq, a.ub, 0, 0, false, 0, a.fast, a.nc_row);
    if (lds > LDS_LIMIT) return c->fail(FH_UNSUPPORTED, "quadrature rule too large for LDS staging");
    if (mode == FH_SCATTER_ATOMIC) {
        a.work_begin = 0;
        a.work_end = (long long)(c->has_mask ? c->num_active : c->E);
        a.labels = c->has_mask ? c->active_list.p : nullptr;
        if (a.work_end == 0) return FH_OK;
        const int grid = (int)((a.work_end + a.epb - 1) / a.epb);
        c->last_kernel = "k_assemble_matrix<atomic>";
#define CALL(EKC, OPC) rc = launch_matrix<EKC, OPC>(c, a, MODE_ATOMIC, lds, grid)
        FH_FOR_ELEM_OP(c->elem_kind, c->op, CALL)
#undef CALL
        return rc;
    }
    if (mode == FH_SCATTER_COLORED) {
        if (!c->has_colors) return c->fail(FH_INVALID_STATE, "FH_SCATTER_COLORED: call fh_color or fh_set_colors first");
        a.labels = c->labels.p;
        c->last_kernel = "k_assemble_matrix<colored>";
        for (size_t col = 0; col + 1 < c->color_offsets.size(); ++col) {
            a.work_begin = (long long)c->color_offsets[col];
            a.work_end = (long long)c->color_offsets[col + 1];
            const long long cntc = a.work_end - a.work_begin;
            if (cntc <= 0) continue;
            const int grid = (int)((cntc + a.epb - 1) / a.epb);
#define CALL(EKC, OPC) rc = launch_matrix<EKC, OPC>(c, a, MODE_COLORED, lds, grid)
            FH_FOR_ELEM_OP(c->elem_kind, c->op, CALL)
#undef CALL
            if (rc) return rc;
        }
        return FH_OK;
    }
    return c->fail(FH_BAD_ARGUMENT, "fh_assemble_matrix: unknown scatter mode");
}

}  // namespace

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
        int rs = c->env_int("FENRIS_HIP_VEC_NT", 256) == 256 ? launch_vector_stream_nt<EK, OP, 256>(c, a) : -1;
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
            static const char* names[9] = {"P0 inputs", "P1 J, inverse", "P2 gradients", "P3 grad u", "P4 F, coefficients", "P5 F^-T g", "MFMA",
                                           "stores", "transposed stores"};
            unsigned long long tot = 0;
            for (int k = 0; k < 9; ++k) tot += h[16 + k];
            for (int k = 0; k < 9; ++k)
                std::fprintf(stderr, "[fenris_hip trace] hex27 %-20s %10.0f cycles/element  %5.1f %%\n", names[k],
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
    if (h > 0 && !c->env("FENRIS_HIP_RECS_LATE") && c->a_recs.n < (size_t)c->E * AFFINE_ROWS_GW_LE)
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

static int apply_mask(fh_ctx* c, const uint8_t* mask);
int fh_set_active_elements(fh_ctx* c, const uint8_t* mask) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, "fh_set_active_elements: set the mesh first");
    c->user_has_mask = mask != nullptr;
    if (mask) c->user_mask.assign(mask, mask + c->E); else c->user_mask.clear();
    return apply_mask(c, mask);
}
static int apply_mask(fh_ctx* c, const uint8_t* mask) {
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
    if (op_kind < FH_LAPLACE || op_kind > FH_MASS_VECTOR) return c->fail(FH_BAD_ARGUMENT, "fh_set_operator: unknown operator");
    if (c->ragged) return c->fail(FH_INVALID_STATE, "fh_set_operator: context holds a ragged connectivity");
    const int old_s = c->S(), old_op = c->op;
    c->op = op_kind;
    if (c->S() != old_s) { c->has_u = false; c->has_tp_pos = false; }
    // the owner-computes partition (LDS budgets, kernel classes, slot parameters) is built for one operator
    if (op_kind != old_op) { c->has_partition = false; ++c->struct_gen; c->has_slotpar = false; c->perm_failed = false; c->rows_try = 0; }
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
static int rs_stage(fh_ctx* c, int g) {
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

// fn(first) once per group that has active elements, with the group staged; restores the caller's element mask
extern "C++" {
template <class F>
static int rs_for_each_group(fh_ctx* c, F&& fn) {
    auto& rs = c->rs;
    std::vector<uint64_t> count(rs.groups.size(), 0);
    for (uint64_t el = 0; el < c->E; ++el)
        if (!c->user_has_mask || c->user_mask[el]) ++count[(size_t)rs.rule_group[rs.e2r[el]]];
    int rc = FH_OK;
    bool first = true;
    for (size_t g = 0; g < rs.groups.size() && rc == FH_OK; ++g) {
        if (count[g] == 0) continue;
        rc = rs_stage(c, (int)g);
        if (rc == FH_OK) rc = fn(first);
        first = false;
    }
    const int rc2 = apply_mask(c, c->user_has_mask ? c->user_mask.data() : nullptr);
    return rc ? rc : rc2;
}
}  // extern "C++"

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
static bool mode_is_colored(int flags) { return (flags & FH_SCATTER_MASK) == FH_SCATTER_COLORED; }
int fh_assemble_matrix_async_dev(fh_ctx* c, double* values_dev, int flags) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->rs.active) return assemble_matrix_enqueue(c, values_dev, flags);
    // rule-set table: one pass per group of rules, the first one with the caller's flags, the others accumulating
    int rc = check_ready(c, "fh_assemble_matrix", true);
    if (rc) return rc;
    if (!values_dev) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_matrix: values is null");
    rc = reset_status(c);
    if (rc) return rc;
    bool any = false;
    rc = rs_for_each_group(c, [&](bool first) {
        any = true;
        if (mode_is_colored(flags) && !c->has_colors) return c->fail(FH_INVALID_STATE, "fh_assemble_matrix: FH_SCATTER_COLORED needs fh_color / fh_set_colors");
        return assemble_matrix_enqueue(c, values_dev, first ? flags : (flags & ~FH_ASSEMBLE_OVERWRITE), false);
    });
    if (rc) return rc;
    if (!any && (flags & FH_ASSEMBLE_OVERWRITE) && fh_nnz(c))
        HIP_TRY(c, hipMemsetAsync(values_dev, 0, sizeof(double) * fh_nnz(c), c->stream));
    return FH_OK;
}

// ---- placement of the streamed buffers (round 3).  The time of the owner-computes kernels follows how the large buffers they stream
// through happen to be backed by device memory -- the same context, kernel and arguments run in one of two or three levels up to 10 %
// apart depending only on WHICH physical memory a buffer got (re-allocating a buffer at the same virtual address changes the level;
// profiles/r03_affine_experiments.txt, section 7).  Nothing in the HIP API chooses the backing, so the library offers the only remedy
// there is: time the real assembly and keep the better of several allocations.
int fh_time_assembly_dev(fh_ctx* c, double* values_dev, int flags, int reps, double* ms_per_assembly) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!ms_per_assembly || reps < 1) return c->fail(FH_BAD_ARGUMENT, "fh_time_assembly_dev: bad argument");
    // reps + 1 REAL assemblies run into the caller's array: without FH_ASSEMBLE_OVERWRITE they would pile up reps + 1 copies of K
    if (!(flags & FH_ASSEMBLE_OVERWRITE)) return c->fail(FH_BAD_ARGUMENT, "fh_time_assembly_dev: needs FH_ASSEMBLE_OVERWRITE (the timed assemblies write the values)");
    int rc = fh_assemble_matrix_async_dev(c, values_dev, flags);   // tables, code objects, first touch
    if (rc) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    if (he == hipSuccess) he = hipEventRecord(e0, c->stream);
    if (he != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        return c->hip_fail(he, "fh_time_assembly_dev");
    }
    for (int k = 0; k < reps && rc == FH_OK; ++k) rc = fh_assemble_matrix_async_dev(c, values_dev, flags);
    he = hipEventRecord(e1, c->stream);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (he != hipSuccess) return c->hip_fail(he, "fh_time_assembly_dev");
    *ms_per_assembly = (double)ms / reps;
    uint64_t failed = 0;
    return fh_poll_status(c, &failed);
}

int fh_tune_placement_dev(fh_ctx* c, double* values_dev, int flags, int tries, double* ms_before, double* ms_after) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    if (!(flags & FH_ASSEMBLE_OVERWRITE)) return c->fail(FH_BAD_ARGUMENT, "fh_tune_placement_dev: needs FH_ASSEMBLE_OVERWRITE (the trial assemblies write the values)");
    double best = 0.0;
    int rc = fh_time_assembly_dev(c, values_dev, flags, 3, &best);
    if (rc) return rc;
    if (ms_before) *ms_before = best;
    if (ms_after) *ms_after = best;
    // the one large buffer of its own that the affine kernels stream through: the element records.  (Moving the position tables and
    // the lane tables never changed the level.)  Rejected allocations are held until the end: freed at once they would be handed out again.
    if (!c->a_recs.p || c->a_npos == 0 || tries < 1) return FH_OK;
    std::vector<double*> rejected;
    const size_t bytes = c->a_recs.n * sizeof(double);
    for (int k = 0; k < tries; ++k) {
        double* cand = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&cand), bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        double* old = c->a_recs.p;
        c->a_recs.p = cand;   // the records are rewritten by every assembly: nothing to copy
        double t = 0.0;
        rc = fh_time_assembly_dev(c, values_dev, flags, 3, &t);
        if (rc == FH_OK && t < 0.98 * best) {
            best = t;
            rejected.push_back(old);
        } else {
            c->a_recs.p = old;
            rejected.push_back(cand);
        }
        if (rc) break;
    }
    (void)hipStreamSynchronize(c->stream);
    for (double* p : rejected) (void)hipFree(p);
    if (ms_after) *ms_after = best;
    return rc;
}

// The rows of the nodes [node_begin, node_end) with a second set of owner-computes tables; the context's own row range and
// tables are untouched.  The second set is built on first use and rebuilt when the range or anything the tables depend on
// (mesh, pattern, mask, operator, quadrature, affine classes) has changed since.
int fh_assemble_matrix_rows_async_dev(fh_ctx* c, double* values_dev, int flags, uint64_t node_begin, uint64_t node_end) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->has_mesh || c->ragged) return c->fail(FH_INVALID_STATE, "fh_assemble_matrix_rows: set the mesh first");
    if (node_begin > node_end || node_end > c->N) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_matrix_rows: bad node range");
    if ((flags & FH_SCATTER_MASK) != FH_SCATTER_GATHER) return c->fail(FH_UNSUPPORTED, "fh_assemble_matrix_rows: needs FH_SCATTER_GATHER");
    if (c->rs.active) return c->fail(FH_UNSUPPORTED, "fh_assemble_matrix_rows: not with a rule-set quadrature table");
    if (!c->rows_stash) c->rows_stash = new PartStash();
    PartStash& st = *c->rows_stash;
    swap_partition(c, st);   // the context's own tables wait in the stash
    if (st.built_gen != c->struct_gen || c->row_lo != (long long)node_begin || c->row_hi != (long long)node_end) {
        c->row_lo = (long long)node_begin;   // a range, even when it covers every node: the two-pass path does not apply
        c->row_hi = (long long)node_end;
        c->has_partition = false;
        c->aff_failed = false;
    }
    c->status_slot = 1;
    const int rc = assemble_matrix_enqueue(c, values_dev, flags);
    c->status_slot = 0;
    swap_partition(c, st);
    st.built_gen = rc ? ~0ull : c->struct_gen;
    return rc;
}

int fh_assemble_matrix_rows_dev(fh_ctx* c, double* values_dev, int flags, uint64_t node_begin, uint64_t node_end, uint64_t* failed) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    const int rc = fh_assemble_matrix_rows_async_dev(c, values_dev, flags, node_begin, node_end);
    if (rc) return rc;
    return read_status(c, failed);
}

int fh_poll_status(fh_ctx* c, uint64_t* failed) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    return read_status(c, failed);
}

int fh_assemble_matrix_dev(fh_ctx* c, double* values_dev, int flags, uint64_t* failed) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = fh_assemble_matrix_async_dev(c, values_dev, flags);
    if (rc) return rc;
    return read_status(c, failed);
}

int fh_assemble_matrix(fh_ctx* c, double* values, int flags, uint64_t* failed) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = check_ready(c, "fh_assemble_matrix", true);
    if (rc) return rc;
    if (!values) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_matrix: values is null");
    const uint64_t nnz = fh_nnz(c);
    DevBuf<double> d;
    HIP_TRY(c, d.alloc((size_t)nnz));
    // the staging copy starts from the caller's values unless every entry is about to be overwritten: with a row range
    // (fh_set_row_range) FH_ASSEMBLE_OVERWRITE writes the rows in range only, "the others are left untouched"
    if (!(flags & FH_ASSEMBLE_OVERWRITE) || c->row_hi >= 0)
        HIP_TRY(c, hipMemcpyAsync(d.p, values, sizeof(double) * nnz, hipMemcpyHostToDevice, c->stream));
    rc = fh_assemble_matrix_dev(c, d.p, flags, failed);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(values, d.p, sizeof(double) * nnz, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FH_OK;
}

int fh_assemble_element_matrices_dev(fh_ctx* c, uint64_t first, uint64_t count, double* ke_dev) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = check_ready(c, "fh_assemble_element_matrices", false);
    if (rc) return rc;
    if (first + count > c->E || (count && !ke_dev)) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_element_matrices: bad range");
    if (count == 0) return FH_OK;
    rc = reset_status(c);
    if (rc) return rc;
    if (c->rs.active) {  // rule-set table: runs of consecutive elements whose rules share points and weights
        const size_t ld = (size_t)c->S() * c->ei.n;
        for (uint64_t e0 = first; e0 < first + count && rc == FH_OK;) {
            const int g = c->rs.rule_group[c->rs.e2r[e0]];
            uint64_t e1 = e0 + 1;
            while (e1 < first + count && c->rs.rule_group[c->rs.e2r[e1]] == g) ++e1;
            if (c->rs.staged != g) rc = rs_stage(c, g);
            if (rc == FH_OK) rc = element_matrices_enqueue(c, e0, e1 - e0, ke_dev + ld * ld * (e0 - first), false);
            e0 = e1;
        }
        const int rc2 = apply_mask(c, c->user_has_mask ? c->user_mask.data() : nullptr);
        if (rc || rc2) return rc ? rc : rc2;
        return read_status(c, nullptr);
    }
    rc = element_matrices_enqueue(c, first, count, ke_dev, false);
    if (rc) return rc;
    return read_status(c, nullptr);
}

int fh_assemble_element_matrices(fh_ctx* c, uint64_t first, uint64_t count, double* ke_out) {
    if (!c) return FH_BAD_ARGUMENT;
    DevGuard dev_guard_(c->device);
    int rc = check_ready(c, "fh_assemble_element_matrices", false);
    if (rc) return rc;
    if (first + count > c->E || (count && !ke_out)) return c->fail(FH_BAD_ARGUMENT, "fh_assemble_element_matrices: bad range");
    if (count == 0) return FH_OK;
    const size_t ld = (size_t)c->S() * c->ei.n;
    DevBuf<double> d;
    HIP_TRY(c, d.alloc(ld * ld * count));
    rc = fh_assemble_element_matrices_dev(c, first, count, d.p);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(ke_out, d.p, sizeof(double) * ld * ld * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FH_OK;
}

// the groups of a rule-set table one after the other; every pass accumulates.  The lowest failing element over all groups
// is reported, like the serial loop of the reference would (global.rs:154: first error aborts).
extern "C++" {
template <class F>
static int rs_walk_accumulating(fh_ctx* c, uint64_t* failed, F&& single) {
    uint64_t fmin = ~0ull;
    bool singular = false;
    int rc = rs_for_each_group(c, [&](bool) {
        uint64_t f = 0;
        const int r = single(&f);
        if (r == FH_SINGULAR_JACOBIAN) { singular = true; fmin = std::min(fmin, f); return (int)FH_OK; }
        return r;
    });
    if (rc) return rc;
    if (singular) {
        if (failed) *failed = fmin;
        return c->fail(FH_SINGULAR_JACOBIAN, "Singular element Jacobian encountered");
    }
    return FH_OK;
}
}  // extern "C++"

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
    if (!a.labels && !c->env("FENRIS_HIP_NO_VECTOR_STREAM")) {
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
static int source_ready(fh_ctx* c, const char* who) {
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
    if (element_pass_covers(c) && !c->env("FENRIS_HIP_NO_VECTOR_TILES") && c->env_int("FENRIS_HIP_ENERGY_TILES", 1)) {
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

static int spmv_launch(fh_ctx* c, const double* vals, const double* x, double* y, double* partial, int grid) {
    const int N = (int)c->N;
    if (c->max_row <= 32 && !c->env("FENRIS_HIP_SPMV_WAVE_PER_NODE")) {   // half a wavefront per node, one lane per column block
        const int un = c->env_int("FENRIS_HIP_SPMV_PAIRS", 2);
        switch (c->S()) {
            case 1: hipLaunchKernelGGL((k_spmv_blocked_half<1>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, partial); break;
            case 2: hipLaunchKernelGGL((k_spmv_blocked_half<2>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, partial); break;
            default:
                if (un >= 4) hipLaunchKernelGGL((k_spmv_blocked_half<3, 4>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, partial);
                else if (un <= 1) hipLaunchKernelGGL((k_spmv_blocked_half<3, 1>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, partial);
                else hipLaunchKernelGGL((k_spmv_blocked_half<3>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, partial);
                break;
        }
        HIP_TRY(c, hipGetLastError());
        return FH_OK;
    }
    switch (c->S()) {
        case 1: hipLaunchKernelGGL((k_spmv_blocked<1>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, partial); break;
        case 2: hipLaunchKernelGGL((k_spmv_blocked<2>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, partial); break;
        default: hipLaunchKernelGGL((k_spmv_blocked<3>), dim3(grid), dim3(256), 0, c->stream, N, c->noff.p, c->ncols.p, vals, x, y, partial); break;
    }
    HIP_TRY(c, hipGetLastError());
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
    const int grid = (int)std::min<uint64_t>(4096, (c->N + 3) / 4);
    c->last_kernel = (c->max_row <= 32 && !c->env("FENRIS_HIP_SPMV_WAVE_PER_NODE")) ? "k_spmv_blocked_half" : "k_spmv_blocked";
    return spmv_launch(c, values_dev, x_dev, y_dev, nullptr, grid);
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
    const int gv = std::min(1024, (n + 255) / 256);                               // vector kernels
    const int gs = (int)std::min<uint64_t>(2048, (c->N + 3) / 4);                  // SpMV: one wavefront per node
    DevBuf<double> r, z, p, Ap, dinv, partial;
    HIP_TRY(c, r.alloc(n));
    HIP_TRY(c, z.alloc(n));
    HIP_TRY(c, p.alloc(n));
    HIP_TRY(c, Ap.alloc(n));
    HIP_TRY(c, partial.alloc((size_t)3 * std::max(gv, gs)));
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
    rc = spmv_launch(c, values_dev, x_dev, r.p, nullptr, gs);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(256), 0, c->stream, n, b_dev, dinv.p, r.p, z.p, p.p, partial.p);
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
        rc = spmv_launch(c, values_dev, p.p, Ap.p, partial.p, gs);
        if (rc) return rc;
        rc = sum_partials(c, partial.p, gs, 1, &pAp);
        if (rc) return rc;
        if (pAp <= 0.0) { status = FH_CG_INDEFINITE_OPERATOR; break; }
        if (zTr <= 0.0) { status = FH_CG_INDEFINITE_PRECONDITIONER; break; }
        const double alpha = zTr / pAp;
        hipLaunchKernelGGL(k_cg_update, dim3(gv), dim3(256), 0, c->stream, n, alpha, p.p, Ap.p, dinv.p, x_dev, r.p, z.p, partial.p);
        HIP_TRY(c, hipGetLastError());
        ++it;
        double s2[2];
        rc = sum_partials(c, partial.p, gv, 2, s2);
        if (rc) return rc;
        const double beta = s2[0] / zTr;
        r_norm = std::sqrt(s2[1]);
        hipLaunchKernelGGL(k_cg_direction, dim3(gv), dim3(256), 0, c->stream, n, beta, z.p, p.p);
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
