// Dense element matrices of tri-quadratic hexahedra (Hex27, s = 3) on the fp64 matrix cores -- round 6 form of the first pass of the two-pass
// owner-computes assembly (LinearElastic / NeoHookean, uniform quadrature table of 27 points).
//
// Per quadrature point the stress contraction of both materials has the form (fenris-solid/src/materials.rs:108-118, 302-313)
//     C(I, J)[i][j] = c_l a_I[i] a_J[j]  -  c_a a_J[i] a_I[j]  +  delta_ij c_m g_I . g_J
// with g_n the physical gradients of the basis, a_n = F^-T g_n (NeoHookean) or a_n = g_n (LinearElastic) and
//     NeoHookean:     c_l = s lambda,  c_a = s (-mu + lambda ln det F),  c_m = s mu      (s = w |det J|, elliptic.rs:422)
//     LinearElastic:  c_l = s lambda,  c_a = -s mu,                      c_m = s mu.
// What this form does differently from hex27_mfma.hpp (rounds 2 - 5; 16 x 16 tiles, G and A = F^-T G both in LDS, eight barriers per element):
//   * ONE operand array.  g = F^T a, so the trace term is  g_I . g_J = a_I^T (F F^T) a_J:  with B = F F^T (3 x 3, symmetric, per point) the row
//     operand of the trace products is  h_d(I) = sum_c (c_m B_cd) a_c(I)  -- nine vector FMAs per fetched row block -- and its column operand
//     is A again.  G is never stored: a_n = M^T r_n with M = J^-1 F^-1 and r_n the REFERENCE gradient, in one phase.  LDS per workgroup
//     52.9 -> 29.8 KB, operand fetches per k-step and wavefront 24 -> 12.
//   * the gradient of u without G: grad u = (sum_n u_n r_n^T) J^-1 -- the sum over the nodes needs no geometry and is formed next to J.
//   * the 3 x 3 algebra of a point (J, its inverse, F, its inverse, M, B, the coefficients) runs on nine lanes of ONE wavefront (seven points per
//     wavefront); the lanes exchange their entries through LDS with no workgroup barrier in between (a wavefront's LDS operations execute
//     in order).  Barriers per element: 3 instead of 8.
//   * the products as 4 x 4 x 4 blocks, v_mfma_f64_4x4x4_4b (75 TFLOP/s on this part against 47 for the 16 x 16 x 4 instruction:
//     profiles/r05_c4_mfma_blocks.txt), row broadcast x column window as in the round-5 experiment.
//   * 128 registers, FOUR workgroups per CU.
// The planar dense layout ke[e][I][i][j][J] is the one k_rows_from_dense<PLANAR> reads.  K_e is symmetric bit for bit: of the symmetric
// components only the blocks on and above the diagonal are multiplied, their mirror images are stored copies (util.rs:38-51 mirrors the upper
// triangle scalar by scalar); K_ji = K_ij^T is a stored copy as well.
#pragma once
#include <hip/hip_runtime.h>

#include "assemble_kernels.hpp"
#include "device_common.hpp"

namespace fenris_hip {

struct Hex27BlkLds {
    // RP rows per component: the 27 nodes and ONE row of zeros (the padding node of the last block); QS: point stride (odd: bank spread), points
    // 27 (zero: the padding of the seventh k-step) and 28 (never read)
    static constexpr int N = 27, NG = 8, NQ = 27, RP = 28, QS = 29;
    static constexpr int o_A = 0;                           // [c][RP][QS]   a_n = M^T r_n
    static constexpr int o_coef = o_A + 3 * RP * QS;        // [9][28]       c_l, -c_a, c_l - c_a, c_m B_00, B_01, B_02, B_11, B_12, B_22 (entry 27 = 0)
    static constexpr int o_M = o_coef + 9 * 28;             // [28][9]       M = J^-1 F^-1, row-major
    static constexpr int o_ggeom = o_M + 28 * 9;            // [q][g][3]     reference gradients of the geometry map
    static constexpr int o_qw = o_ggeom + NQ * NG * 3;      // [28]
    static constexpr int o_X = o_qw + 28;                   // [g][3]
    static constexpr int o_U = o_X + NG * 3;                // [n][3]
    static constexpr int total = o_U + N * 3 + 1;
    static constexpr int KE_TRI = (N * (N + 1) / 2) * 9;    // doubles per element in memory: the upper node-block triangle
    // scratch of the per-point chain: inside A, which is dead between the matrix phase of one element and phase P2 of the next.  Only rows
    // 0 .. 26 of a component are used (the row of zeros stays); the zeros of point 27 in those rows are rewritten by P2.
    static constexpr int s_P = 0;                           // [q][27]  partial sums of sum_n u_n r_n^T: [lane group][c][m]
    static constexpr int s_J = RP * QS;                     // [q][9]   J, later F
    static constexpr int s_I = s_J + 244;                   // [q][9]   J^-1
    static constexpr int s_H = 2 * RP * QS;                 // [q][9]   sum_n u_n r_n^T, later F^-1
    static_assert(s_P + 27 * 27 <= 27 * QS && s_I + 244 <= RP * QS + 27 * QS && s_H + 244 <= 2 * RP * QS + 27 * QS, "scratch inside rows 0 .. 26");
};

// log() out of line (the compiler otherwise keeps its polynomial's coefficients in registers across the element loop)
static __device__ __attribute__((noinline)) double hex27b_log(double x) { return log(x); }

template <int OP, bool TRACE = false>
__global__ void __launch_bounds__(256, 4) k_hex27_dense_blocks(const KArgs a, double mu_u, double lambda_u) {
    using L = Hex27BlkLds;
    constexpr int N = L::N, NG = L::NG, NQ = L::NQ, RP = L::RP, QS = L::QS;
    constexpr bool NH = (OP == FH_NEO_HOOKEAN);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, nt = 256;
    for (int i = tid; i < NQ * NG * 3; i += nt) lds[L::o_ggeom + i] = a.ggeom[i];
    for (int i = tid; i < 28; i += nt) lds[L::o_qw + i] = (i < NQ) ? a.qw[i] : 0.0;
    for (int i = tid; i < L::o_ggeom; i += nt) lds[i] = 0.0;   // A (with its row and its point of zeros), coefficients, M
    __syncthreads();
    double* A = lds + L::o_A;
    // FENRIS_HIP_TRACE: cycles of wavefront 0 per phase (P0 + barrier, chain, barrier, P2 + barrier, matrix phase, stores), summed over
    // workgroups into trace[16 + phase]; trace[31] counts elements
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_t = 0, ph_n = 0;
    const bool tracing = TRACE && a.trace != nullptr;
    auto mark = [&](int k) {
        if (tracing) { const unsigned long long t = __builtin_readcyclecounter(); ph[k] += t - ph_t; ph_t = t; }
    };

    // The element's inputs -- 8 geometry vertices, 27 values of u: element index, node index, then the gather -- are prefetched in registers
    // as a chain of three requests, each consumed one element after it was issued (hex27_mfma.hpp, unchanged): EVERY thread takes part and
    // nothing is fetched under a branch.
    // (Everything that depends on the thread id alone -- roles, table offsets -- is formed anew per element from an opaque copy of the id: kept in
    // registers across the matrix phase it would take a dozen of the 128 there are.)
    struct Roles { bool xrole, urole, val_used; int ri; const double* src; };
    auto roles_of = [&](int t) {
        Roles r;
        r.xrole = t < NG * 3;
        r.urole = NH && t >= 64 && t < 64 + N * 3;
        r.ri = r.xrole ? t : (r.urole ? t - 64 : 0);
        r.src = (r.urole && a.u) ? a.u : a.verts;
        r.val_used = r.xrole || (r.urole && a.u);
        return r;
    };
    const long long Gs = gridDim.x;
    auto elem_of = [&](long long w) { const long long wc = min(w, a.work_end - 1); return a.labels ? (long long)a.labels[wc] : wc; };
    // (the connectivity and the gradient tables come with the nodes in lexicographic order of their reference positions -- engine_two_pass.hip --,
    // geometry vertex g sits at place (vtx_pack >> 5 g) & 31)
    auto node_at = [&](const Roles& r, long long e) {
        const int nd = r.ri / 3;
        return a.conn[(size_t)e * N + (r.xrole ? (int)((a.vtx_pack >> (5 * nd)) & 31ull) : nd)];
    };
    auto value_at = [&](const Roles& r, int node) { return r.src[(size_t)node * 3 + r.ri % 3]; };
    const long long w0 = a.work_begin + blockIdx.x;
    if (w0 >= a.work_end) return;
    long long e_cur = elem_of(w0), e_n1 = elem_of(w0 + Gs), e_n2 = elem_of(w0 + 2 * Gs);
    double val_cur;
    int node_n1;
    {
        const Roles r = roles_of(tid);
        val_cur = value_at(r, node_at(r, e_cur));
        node_n1 = node_at(r, e_n1);
    }

    // lane roles of the per-point chain: nine lanes per point, seven points per wavefront (lane 63 and the seventh group of wavefront 3 idle)
    struct Chain { int pe, pi, pj, pq; bool valid; };
    auto chain_of = [&](int t) {
        Chain c;
        const int ln = t & 63, pl = ln / 9;
        c.pe = ln - 9 * pl; c.pi = c.pe / 3; c.pj = c.pe - 3 * c.pi;
        const int q_raw = 7 * (t >> 6) + pl;
        c.valid = pl < 7 && q_raw < NQ;
        c.pq = min(q_raw, NQ - 1);
        return c;
    };

    // reference gradients, the same for every element but fetched anew for each -- ahead of the previous element's stores, so that the wait
    // for them is not a wait for those stores (loads and stores share one in-order counter):
    //   gr[k][.]: this thread's three (point, node) items of phase P2 (item = tid + 256 k; beyond the table: zeros by the bounds check);
    //   r9[j]:    component pj of node 3 j + pi at the lane's point (chain lanes; from the node-major copy of the table: the 21 values of a
    //             wavefront's seven points are one run)
    // Buffer loads: one offset register per table, the rest of the address is a scalar / immediate.
    double gr[3][3], r9[9];
    const auto gref_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.gref), (short)0, NQ * N * 3 * 8, 0x00020000);
    const auto greft_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(NH ? a.gref_t : a.gref), (short)0, NQ * N * 3 * 8, 0x00020000);
    auto load_gref = [&]() {
        typedef unsigned ld_u32x2 __attribute__((ext_vector_type(2)));
        int t = tid;
        asm volatile("" : "+v"(t));
        const unsigned vo = (unsigned)t * 24u;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
                gr[k][cc] = __builtin_bit_cast(double, (ld_u32x2)__builtin_amdgcn_raw_buffer_load_b64(gref_rsrc, vo + (unsigned)(cc * 8), k * nt * 24, 0));
        if (NH) {
            const Chain c = chain_of(t);
            const unsigned vt = (unsigned)((c.pi * NQ + c.pq) * 24 + c.pj * 8);
#pragma unroll
            for (int j = 0; j < 9; ++j)
                r9[j] = __builtin_bit_cast(double, (ld_u32x2)__builtin_amdgcn_raw_buffer_load_b64(greft_rsrc, vt, j * 3 * NQ * 24, 0));
        }
    };
    load_gref();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing pending at the loop's entry (see hex27_mfma.hpp)
    for (long long w = w0; w < a.work_end; w += Gs) {
        const long long e = e_cur;
        if (tracing) { ph_t = __builtin_readcyclecounter(); ++ph_n; }
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const Roles ro = roles_of(t_);
        const Chain ch = chain_of(t_);
        const int pe = ch.pe, pi = ch.pi, pj = ch.pj, pq = ch.pq;
        const bool pvalid = ch.valid;
        // ---- P0: geometry nodes (the first 8) and u of the element, from the registers; requests for the next elements.  The barrier is also
        // the end of the previous element's matrix phase: A, the coefficients and M are free behind it.
        if (ro.xrole) lds[L::o_X + ro.ri] = val_cur;
        if (ro.urole) lds[L::o_U + ro.ri] = ro.val_used ? val_cur : 0.0;
        double val_n1 = value_at(ro, node_n1);
        int node_n2 = node_at(ro, e_n2);
        long long e_n3 = elem_of(w + 3 * Gs);
        lds_barrier();
        mark(0);
        // vector fp64 work shares the SIMD's datapath with the matrix instructions of the CU's other workgroups: raised priority for the chain
        __builtin_amdgcn_s_setprio(3);
        // ---- P1: the chain of a point on nine lanes, entry (pi, pj) each; exchanges through LDS inside the wavefront
        if (!(TRACE && (a.ablate & 1))) {
            double* sJ = A + L::s_J + pq * 9;
            double* sI = A + L::s_I + pq * 9;
            double* sH = A + L::s_H + pq * 9;
            double* sP = A + L::s_P + pq * 27;
            // J = X G^T (hexahedron.rs:324-326 -> :101-107)
            double Jv = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) Jv = fma(lds[L::o_X + g * 3 + pi], lds[L::o_ggeom + (pq * NG + g) * 3 + pj], Jv);
            if (pvalid) sJ[pe] = Jv;
            if (NH) {
                // this lane's nodes n = 3 j + pi: partial sums of H[c][m] = sum_n u_n[c] r_n[m] for m = pj, all three c
                double P0 = 0.0, P1 = 0.0, P2 = 0.0;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const double* un = lds + L::o_U + (3 * j + pi) * 3;
                    P0 = fma(un[0], r9[j], P0);
                    P1 = fma(un[1], r9[j], P1);
                    P2 = fma(un[2], r9[j], P2);
                }
                if (pvalid) { sP[pi * 9 + 0 + pj] = P0; sP[pi * 9 + 3 + pj] = P1; sP[pi * 9 + 6 + pj] = P2; }
            }
            asm volatile("" ::: "memory");
            // det J, entry (pi, pj) of J^-1 = cofactor (pj, pi) / det (what nalgebra's try_inverse does; it fails for det == 0 exactly,
            // elliptic.rs:401-404)
            const int r1 = (pj + 1) % 3, r2 = (pj + 2) % 3, c1 = (pi + 1) % 3, c2 = (pi + 2) % 3;
            double detJ;
            {
                const double J[3][3] = {{sJ[0], sJ[1], sJ[2]}, {sJ[3], sJ[4], sJ[5]}, {sJ[6], sJ[7], sJ[8]}};
                detJ = det_small<3>(J);
            }
            double Jiv = 0.0;
            if (detJ == 0.0) {
                if (pvalid && pe == 0) report_singular(a.status, e);
            } else {
                Jiv = (sJ[r1 * 3 + c1] * sJ[r2 * 3 + c2] - sJ[r2 * 3 + c1] * sJ[r1 * 3 + c2]) * (1.0 / detJ);
            }
            const double s = lds[L::o_qw + pq] * fabs(detJ);  // elliptic.rs:422
            const double mu = a.qparams ? a.qparams[2 * pq] : mu_u, lambda = a.qparams ? a.qparams[2 * pq + 1] : lambda_u;
            double Mv = Jiv, Bv = (pi == pj) ? 1.0 : 0.0, c_a = -(s * mu);
            const double c_l = s * lambda, c_m = s * mu;
            if (NH) {
                const double Hv = (sP[0 + pe] + sP[9 + pe]) + sP[18 + pe];   // H[pi][pj]
                asm volatile("" ::: "memory");
                if (pvalid) { sI[pe] = Jiv; sH[pe] = Hv; }
                asm volatile("" ::: "memory");
                // F = I + (grad u)^T (fenris-solid/src/lib.rs:20-29), grad u = sum_n g_n u_n^T with g_n = J^-T r_n:  F = I + H J^-1
                const double Fv = ((pi == pj) ? 1.0 : 0.0) + fma(sH[pi * 3 + 0], sI[0 + pj], fma(sH[pi * 3 + 1], sI[3 + pj], sH[pi * 3 + 2] * sI[6 + pj]));
                asm volatile("" ::: "memory");
                if (pvalid) sJ[pe] = Fv;      // (every lane of the wavefront has read J by now: in-order LDS)
                asm volatile("" ::: "memory");
                double Jd;
                {
                    const double F[3][3] = {{sJ[0], sJ[1], sJ[2]}, {sJ[3], sJ[4], sJ[5]}, {sJ[6], sJ[7], sJ[8]}};
                    Jd = det_small<3>(F);
                }
                // entry of F^-1 (zeros where det F <= 0: the coefficients are NaN there, materials.rs:298-300, and NaN x 0 = NaN)
                double Fiv = 0.0;
                if (Jd > 0.0) Fiv = (sJ[r1 * 3 + c1] * sJ[r2 * 3 + c2] - sJ[r2 * 3 + c1] * sJ[r1 * 3 + c2]) * (1.0 / Jd);
                // B = F F^T
                Bv = fma(sJ[pi * 3 + 0], sJ[pj * 3 + 0], fma(sJ[pi * 3 + 1], sJ[pj * 3 + 1], sJ[pi * 3 + 2] * sJ[pj * 3 + 2]));
                asm volatile("" ::: "memory");
                if (pvalid) sH[pe] = Fiv;     // (H has been read)
                asm volatile("" ::: "memory");
                // M = J^-1 F^-1
                Mv = fma(sI[pi * 3 + 0], sH[0 + pj], fma(sI[pi * 3 + 1], sH[3 + pj], sI[pi * 3 + 2] * sH[6 + pj]));
                if (pe == 0) c_a = (Jd > 0.0) ? s * (-mu + lambda * hex27b_log(Jd)) : __builtin_nan("");
                if (!(Jd > 0.0)) Bv = __builtin_nan("");   // the whole block is NaN (c_l, c_m as well, below)
            }
            if (pvalid) {
                lds[L::o_M + pq * 9 + pe] = Mv;
                if (pi <= pj) lds[L::o_coef + (3 + pe - (pi * (pi + 1)) / 2) * 28 + pq] = c_m * Bv;
                if (pe == 0) {
                    const bool bad = NH && c_a != c_a;
                    const double cl = bad ? c_a : c_l;
                    lds[L::o_coef + 0 * 28 + pq] = cl;
                    lds[L::o_coef + 1 * 28 + pq] = -c_a;
                    lds[L::o_coef + 2 * 28 + pq] = cl - c_a;
                }
            }
        }
        lds_barrier();
        mark(1);
        // ---- P2: one lane per (point, node): a_n = M^T r_n; the zeros of point 27 (the chain's scratch lay over them)
        if (!(TRACE && (a.ablate & 1))) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int it = t_ + nt * k;
                if (it < NQ * N) {
                    const int q = it / N, n = it - q * N;
                    const double* Mq = lds + L::o_M + q * 9;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        A[(c * RP + n) * QS + q] = fma(Mq[0 * 3 + c], gr[k][0], fma(Mq[1 * 3 + c], gr[k][1], Mq[2 * 3 + c] * gr[k][2]));
                } else if (it < NQ * N + N) {
                    const int n = it - NQ * N;
#pragma unroll
                    for (int c = 0; c < 3; ++c) A[(c * RP + n) * QS + NQ] = 0.0;
                }
            }
        }
        lds_barrier();
        mark(2);
        __builtin_amdgcn_s_setprio(0);
        // ---- matrix cores: the 28 node blocks (4 x 4 nodes) ON AND ABOVE the diagonal, every one with all nine components.
        // v_mfma_f64_4x4x4_4b multiplies four independent 4 x 4 x 4 blocks; register layout (scripts/ubench/mfma_f64_4x4_layout.hip):
        // A: lane = (i + 4 g) + 16 k, B: lane = (j + 4 g) + 16 k, D: lane = (j + 4 g) + 16 i, g = the block.  WHICH (row block, column block) a
        // group works on is the lane's choice of LDS address, so the 28 blocks are packed into 7 units of four with nothing wasted:
        //     role 0: (0,0) (0,1) (0,2) (0,3) | (0,4) (0,5) (0,6) (6,6)        role 1: (1,1) (1,2) (1,3) (1,4) | (1,5) (1,6) (5,5) (5,6)
        //     role 2: (2,2) (2,3) (2,4) (2,5) | (2,6) (4,4) (4,5) (4,6)        role 3: (3,3) (3,4) (3,5) (3,6)
        // (roles rotate from element to element: every SIMD gets the light one).  Per unit and k-step: 3 + 3 products for the three K_ii and
        // the trace term, 12 for the six K_ij, i != j:  K_ij(I, J) = (c_l a_I[i]) a_J[j] + (-c_a a_I[j]) a_J[i]  -- the components below the
        // diagonal are multiplied like the ones above it, on the upper node blocks only: the same 18 products per upper block as with mirrored
        // components, but a lane then holds the COMPLETE 3 x 3 block of its node pair (I, J) and stores 72 contiguous bytes; nothing is
        // mirrored.  The element matrix goes to memory as its upper node-block triangle, ke[e][tri(I, J)][i][j], tri(I, J) = I (53 - I) / 2 + J
        // for I <= J (27 216 bytes instead of 52 488); k_rows_from_dense<TRI> reads a block (J, I), J < I, transposed.  126 matrix
        // instructions per k-step and workgroup (the tiles: 42 of four times the size).
        const int role = (__builtin_amdgcn_readfirstlane(t_ >> 6) + (int)(((w - w0) / Gs) & 3)) & 3;
        // row / column block of (unit u, group g) in bits 3 (g + 4 u) ..; 7 = idle (the row of zeros)
        constexpr unsigned IBT[4] = {0u | 0u << 3 | 0u << 6 | 0u << 9 | 0u << 12 | 0u << 15 | 0u << 18 | 6u << 21,
                                     1u | 1u << 3 | 1u << 6 | 1u << 9 | 1u << 12 | 1u << 15 | 5u << 18 | 5u << 21,
                                     2u | 2u << 3 | 2u << 6 | 2u << 9 | 2u << 12 | 4u << 15 | 4u << 18 | 4u << 21,
                                     3u | 3u << 3 | 3u << 6 | 3u << 9 | 7u << 12 | 7u << 15 | 7u << 18 | 7u << 21};
        constexpr unsigned JBT[4] = {0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12 | 5u << 15 | 6u << 18 | 6u << 21,
                                     1u | 2u << 3 | 3u << 6 | 4u << 9 | 5u << 12 | 6u << 15 | 5u << 18 | 6u << 21,
                                     2u | 3u << 3 | 4u << 6 | 5u << 9 | 6u << 12 | 4u << 15 | 5u << 18 | 6u << 21,
                                     3u | 4u << 3 | 5u << 6 | 6u << 9 | 7u << 12 | 7u << 15 | 7u << 18 | 7u << 21};
        const unsigned ibt = role == 0 ? IBT[0] : (role == 1 ? IBT[1] : (role == 2 ? IBT[2] : IBT[3]));
        const unsigned jbt = role == 0 ? JBT[0] : (role == 1 ? JBT[1] : (role == 2 ? JBT[2] : JBT[3]));
        int lane_o = t_ & 63;
        asm volatile("" : "+v"(lane_o));   // (everything derived from it is formed per element instead of living in registers across the loop)
        const int gq = (lane_o >> 2) & 3, idx = lane_o & 3, kq4 = lane_o >> 4;
        auto node_off = [&](unsigned tbl, int u) {   // LDS byte offset of the lane's operand row: node 4 B + (lane & 3) of the group's block B
            const int B = (int)((tbl >> (3 * (gq + 4 * u))) & 7u);
            return (unsigned)((min(4 * B + idx, N) * QS + kq4) * 8);
        };
        const char* Ab = reinterpret_cast<const char*>(A);
        const char* Cb = reinterpret_cast<const char*>(lds + L::o_coef);
        auto ldA = [&](int comp, unsigned off) { return *reinterpret_cast<const double*>(Ab + off + (size_t)(comp * RP * QS * 8)); };
        auto ldC = [&](int k, unsigned off) { return *reinterpret_cast<const double*>(Cb + off + (size_t)(k * 28 * 8)); };
        auto mm = [](double x, double y, double acc) { return __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc, 0, 0, 0); };
        const unsigned long long ke_addr = reinterpret_cast<unsigned long long>(a.ke_out + (size_t)e * L::KE_TRI);
        const unsigned long long ke_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ke_addr >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)ke_addr);
        const auto ke_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ke_u), (short)0, L::KE_TRI * 8, 0x00020000);
        auto rotate_inputs = [&]() {
            // The requests of this element are consumed here: not earlier (they need their time), and not behind the stores below
            asm volatile("" : "+v"(val_n1), "+v"(node_n2), "+v"(e_n3));
            val_cur = val_n1;
            node_n1 = node_n2;
            e_cur = e_n1;
            e_n1 = e_n2;
            e_n2 = e_n3;
        };
        // accumulators of a unit: the lane's 3 x 3 block, the diagonal entries still without the trace term
        struct Unit { double k[3][3], m; };
        auto zero_unit = [](Unit& x) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) x.k[i][j] = 0.0;
            x.m = 0.0;
        };
        // one k-step of a unit: row operands r (scaled here), column operands wb
        auto step_unit = [&](Unit& x, const double (&r)[3], const double (&wb)[3], const double (&cf)[9]) {
            const double s0 = cf[2] * r[0], s1 = cf[2] * r[1], s2 = cf[2] * r[2];
            const double h0 = fma(cf[5], r[2], fma(cf[4], r[1], cf[3] * r[0]));
            const double h1 = fma(cf[7], r[2], fma(cf[6], r[1], cf[4] * r[0]));
            const double h2 = fma(cf[8], r[2], fma(cf[7], r[1], cf[5] * r[0]));
            const double l0 = cf[0] * r[0], l1 = cf[0] * r[1], l2 = cf[0] * r[2];
            const double n0 = cf[1] * r[0], n1 = cf[1] * r[1], n2 = cf[1] * r[2];
            // (consecutive instructions on different accumulators)
            x.k[0][0] = mm(s0, wb[0], x.k[0][0]); x.k[1][1] = mm(s1, wb[1], x.k[1][1]); x.k[2][2] = mm(s2, wb[2], x.k[2][2]);
            x.m = mm(h0, wb[0], x.m);
            x.k[0][1] = mm(l0, wb[1], x.k[0][1]); x.k[0][2] = mm(l0, wb[2], x.k[0][2]); x.k[1][0] = mm(l1, wb[0], x.k[1][0]);
            x.k[1][2] = mm(l1, wb[2], x.k[1][2]); x.k[2][0] = mm(l2, wb[0], x.k[2][0]); x.k[2][1] = mm(l2, wb[1], x.k[2][1]);
            x.m = mm(h1, wb[1], x.m);
            x.k[0][1] = mm(n1, wb[0], x.k[0][1]); x.k[0][2] = mm(n2, wb[0], x.k[0][2]); x.k[1][0] = mm(n0, wb[1], x.k[1][0]);
            x.k[1][2] = mm(n2, wb[1], x.k[1][2]); x.k[2][0] = mm(n0, wb[2], x.k[2][0]); x.k[2][1] = mm(n1, wb[2], x.k[2][1]);
            x.m = mm(h2, wb[2], x.m);
        };
        // the lane's block of unit u to memory: 72 contiguous bytes (four 16-byte stores and one of 8); buffer stores with the hardware's bounds
        // check -- padding nodes, node pairs below the diagonal and idle groups get an offset beyond the element's triangle: dropped
        auto store_unit = [&](Unit& x, int u, int ls) {
            typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
            typedef unsigned st_u32x2 __attribute__((ext_vector_type(2)));
            typedef double st_f64x2 __attribute__((ext_vector_type(2)));
            const int g = (ls >> 2) & 3;
            const int IB = (int)((ibt >> (3 * (g + 4 * u))) & 7u), JB = (int)((jbt >> (3 * (g + 4 * u))) & 7u);
            const int I = 4 * IB + (ls >> 4), J = 4 * JB + (ls & 3);
            const bool valid = I < N && J < N && I <= J;
            const unsigned vo = valid ? (unsigned)((I * (53 - I)) / 2 + J) * 72u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < 3; ++i) x.k[i][i] += x.m;
            if (I == J) {   // a node with itself: symmetric bit for bit (util.rs:38-51 copies the upper triangle)
                x.k[1][0] = x.k[0][1]; x.k[2][0] = x.k[0][2]; x.k[2][1] = x.k[1][2];
            }
            const double* f = &x.k[0][0];
            if (TRACE && (a.ablate & 8)) {   // timing only (wrong places): the same bytes as whole 1 KB runs per instruction -- what ideal stores would cost
                const unsigned base = (unsigned)((role * 2 + u) * 4608);
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) {
                    const st_f64x2 v = {f[2 * p2], f[2 * p2 + 1]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(st_u32x4, v), ke_rsrc, (base + (unsigned)(p2 * 1024 + ls * 16)) % 27200u, 0, 0);
                }
                if (ls < 32) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(st_u32x4, (st_f64x2){f[8], f[8]}), ke_rsrc, (base + (unsigned)(4096 + ls * 16)) % 27200u, 0, 0);
                return;
            }
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                const st_f64x2 v = {f[2 * p2], f[2 * p2 + 1]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(st_u32x4, v), ke_rsrc, vo, 16 * p2, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(st_u32x2, f[8]), ke_rsrc, vo, 64, 0);
        };
        unsigned oC = (unsigned)(kq4 * 8);
        if (role < 3) {
            unsigned oR0 = node_off(ibt, 0), oC0 = node_off(jbt, 0), oR1 = node_off(ibt, 1), oC1 = node_off(jbt, 1);
            Unit x0, x1;
            zero_unit(x0);
            zero_unit(x1);
            if (!(TRACE && (a.ablate & 2))) {
#pragma unroll 1
                for (int ks = 0; ks < 7; ++ks) {
                    double r0[3], c0[3], r1[3], c1[3], cf[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) cf[k] = ldC(k, oC);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { r0[c] = ldA(c, oR0); c0[c] = ldA(c, oC0); }
#pragma unroll
                    for (int c = 0; c < 3; ++c) { r1[c] = ldA(c, oR1); c1[c] = ldA(c, oC1); }
                    step_unit(x0, r0, c0, cf);
                    step_unit(x1, r1, c1, cf);
                    oC += 32u; oR0 += 32u; oC0 += 32u; oR1 += 32u; oC1 += 32u;
                }
            }
            rotate_inputs();
            mark(3);
            load_gref();
            asm volatile("" ::: "memory");
            if (!(TRACE && (a.ablate & 4))) {
                int ls = tid & 63;
                asm volatile("" : "+v"(ls));
                store_unit(x0, 0, ls);
                store_unit(x1, 1, ls);
            }
        } else {
            unsigned oR0 = node_off(ibt, 0), oC0 = node_off(jbt, 0);
            Unit x0;
            zero_unit(x0);
            if (!(TRACE && (a.ablate & 2))) {
#pragma unroll 1
                for (int ks = 0; ks < 7; ++ks) {
                    double r0[3], c0[3], cf[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) cf[k] = ldC(k, oC);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { r0[c] = ldA(c, oR0); c0[c] = ldA(c, oC0); }
                    step_unit(x0, r0, c0, cf);
                    oC += 32u; oR0 += 32u; oC0 += 32u;
                }
            }
            rotate_inputs();
            mark(3);
            load_gref();
            asm volatile("" ::: "memory");
            if (!(TRACE && (a.ablate & 4))) {
                int ls = tid & 63;
                asm volatile("" : "+v"(ls));
                store_unit(x0, 0, ls);
            }
        }
        mark(4);
        // (no barrier here: the next element's P0 writes X and U only, and its barrier stands between this matrix phase and the next chain)
    }
    if (tracing && tid == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(a.trace + 16 + k, ph[k]);
        atomicAdd(a.trace + 31, ph_n);
        a.trace[30] = 2;   // (which kernel's phases these are: engine.hip prints them at fh_destroy)
    }
}

}  // namespace fenris_hip
