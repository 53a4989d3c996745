#pragma once
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
#include "host_pool.hpp"
#include "pattern_kernels.hpp"

extern char** environ;

using namespace fenris_hip;

namespace fenris_hip_detail {

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
inline void ref_gradients(int kind, const double* xi, double* out) {
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
inline void ref_basis(int kind, const double* xi, double* out) {
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
inline bool elem_info(int kind, ElemInfo& e) {
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

}  // namespace fenris_hip_detail
using namespace fenris_hip_detail;

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
    DevBuf<double> gref_t;      // Hex27: reference gradients node-major (KArgs::gref_t)
    // Hex27 matrix-core first pass: the element's nodes in LEXICOGRAPHIC order of their reference positions (x fastest) instead of the element's own
    // (vertices, edges, faces, centre): neighbouring nodes of a mesh row then own neighbouring blocks of the stored triangle, and the second pass
    // finds the lines it shares with its neighbours in the L2 / L1 (engine_two_pass.hip).  perm[n] = place of local node n.
    DevBuf<double> gref_lex, gref_t_lex;   // the two gradient tables with their node index permuted
    int hex27_perm[27] = {0};
    unsigned long long hex27_vtx_pack = 0;  // 5 bits per geometry vertex: its place
    bool has_hex27_perm = false;
    DevBuf<double> tensor;      // FH_TENSOR: nq x d^4 coefficient tensors (fh_set_operator_tensor); tensor_nq = the point count they were given for
    int tensor_nq = 0;
    bool tensor_sym = true;
    DevBuf<double> qmono;       // Hex8: coordinates and pair products of the quadrature points (KArgs::qmono)
    DevBuf<double> qmom;        // Hex8: moments of the rule (KArgs::qmom); qmom_ok: the rule is symmetric and the parameters are the same at every point
    bool qmom_ok = false;
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
    int rows_try = 0;           // block sizes tried for them: 13 nodes / 352 entries, 11 / 288, 9 / 256, 7 / 224, then the standard form
    DevBuf<unsigned> r_lanes4;  //                                     lanes per position (Tet4): three words each
    DevBuf<int> r_vconn;        //                                     unique vertices + slot words per position (Tet4)
    int r_rw = 0, r_ls = 256, r_vn = 256;   // (r_vn: vertices per position in r_vconn)
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
    DevBuf<uint2> a_lanes;          // lane records
    DevBuf<int4> a_hdr;             // position headers
    DevBuf<double> a_recs;          // element records (R or M), rewritten by every assembly
    int a_us = 0, a_npos = 0, a_ntab = 0, a_incomplete = 0;
    // general Hex8 row-owner kernel (hex8_rows.hip): lane tables and position records of the GENERAL positions (p_rec order)
    DevBuf<int4> h_hdr, h_pos;
    DevBuf<uint2> h_lanes;
    int h_ntab = 0, h_incomplete = 0;
    int h_tune_pending = 0;   // k_hex8_rows: launches left before the lane tuner runs (0: done or off), see hex8_tune_lanes_now
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
    unsigned long long pattern_gen = 0;      // counts pattern builds: what caches device pointers of the pattern (group.hip) compares
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
    int tp_pos_layout = -1;            // the layout (engine_two_pass.hip) the cached tables were built for
    DevBuf<int> tp_conn;               // triangle layout: connectivity with permuted columns, and the (node, element) entries with the permuted local index
    DevBuf<unsigned> tp_adj;
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
    X(blk_off) X(gt_elems) X(gt_ent) X(gt_pos) X(has_pos) X(p_conn) X(p_rec) X(p_elem) X(r_rec) X(r_lanes4) X(r_vconn) X(r_rw) X(r_ls) X(r_vn)  \
    X(has_rows) X(p_rw) X(p_cs) X(p_ms) X(p_nbs) X(p_jt) X(p_us) X(has_pipe) X(gt_hdr) X(nblk) X(g_ub) X(g_mb) X(g_acc)      \
    X(g_nb) X(g_umax) X(has_partition) X(a_conn) X(a_elem) X(a_lanes) X(a_hdr) X(a_us) X(a_npos) X(a_ntab) X(a_incomplete) X(a_emin) X(a_emax) X(npos_gen)       \
    X(h_hdr) X(h_pos) X(h_lanes) X(h_ntab) X(h_incomplete) X(h_tune_pending) X(has_hrows) X(aff_failed) X(row_lo) X(row_hi) X(p_slotpar) X(has_slotpar) X(part_perm) X(part_rows_only) X(rows_try) X(perm_failed)
struct PartStash {
#define X(name) decltype(fh_ctx::name) name{};
    FH_PARTITION_MEMBERS(X)
#undef X
    unsigned long long built_gen = ~0ull;
    PartStash() { row_lo = 0; row_hi = -1; r_ls = 256; r_vn = 256; p_jt = 1; }
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


// ---- functions shared by the translation units of the engine (engine.hip: context, mesh, pattern, options; engine_partition.hip: owner
// blocks and position tables; engine_matrix.hip: stiffness / mass launchers; engine_two_pass.hip: dense element matrices + row gather;
// engine_vector.hip: residual, source, energy; engine_solver.hip: Dirichlet rows, SpMV, PCG, error integrals)
constexpr size_t LDS_TARGET = 64 * 1024;   // two workgroups per CU
constexpr size_t LDS_LIMIT = 160 * 1024;   // hardware limit per workgroup
inline bool generic_fast(const fh_ctx* c) { return c->fast_ok && (!c->has_rules || c->op == FH_LAPLACE); }
int grid_for(long long n, int block, int cap = 256 * 32);
int source_ready(fh_ctx* c, const char* who);
void invalidate_pattern(fh_ctx* c);
int build_compute_adjacency(fh_ctx* c);
int build_source_adjacency(fh_ctx* c);
int host_offsets(fh_ctx* c);
int build_pattern(fh_ctx* c);
int check_ready(fh_ctx* c, const char* who, bool need_pattern);
void fill_common(fh_ctx* c, KArgs& a);
int reset_status(fh_ctx* c);
int read_status(fh_ctx* c, uint64_t* failed);
int choose_epb(fh_ctx* c, int what, size_t lds_target = LDS_TARGET);
int build_partition(fh_ctx* c);
size_t layout_bytes_dyn(int ek, int op, int what, int nq, int ub, int acc, int nb, bool gather, int mb = 0, int fast = 0, int nc_row = 0);
int element_matrices_enqueue(fh_ctx* c, uint64_t first, uint64_t count, double* ke_dev, bool by_elem, bool tri = false);
int assemble_two_pass(fh_ctx* c, double* values_dev, int overwrite);
size_t two_pass_dense_doubles(fh_ctx* c);   // doubles of the element-matrix buffer between the two passes (depends on the first pass's layout)
int hex8_tune_lanes_now(fh_ctx* c);
int assemble_matrix_enqueue(fh_ctx* c, double* values_dev, int flags, bool reset = true);

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
        case FH_TENSOR: CALL(EKC, FH_TENSOR); break;                \
        default: break;                                             \
    }

// ---- rule-set tables: the groups of rules that share points and weights, one pass each
int rs_stage(fh_ctx* c, int g);
int apply_mask(fh_ctx* c, const uint8_t* mask);
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
