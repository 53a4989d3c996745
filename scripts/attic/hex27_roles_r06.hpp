// Dense element matrices of Hex27 on the fp64 matrix cores, WAVE-SPECIALISED form of hex27_blocks.hpp (round 6, second half).
//
// In k_hex27_dense_blocks every wavefront alternates between the element's prologue (per-point 3 x 3 chain, a = M^T r) and its matrix phase, with
// three barriers per element; per element and workgroup it spends ~30 k cycles of which 11.6 k in the matrix phase, and the matrix cores are 43 %
// busy (profiles/r06_c4_triangle.txt).  Here a workgroup has EIGHT wavefronts with fixed roles:
//   * wavefronts 0 - 3: matrix phase only -- the 7 units of upper node blocks, 18 products per block and k-step, 72 contiguous bytes per lane out
//     (exactly the matrix phase of hex27_blocks.hpp), element after element;
//   * wavefronts 4 - 7: the prologue of the NEXT element into the other half of a double-buffered operand array: each takes seven of the 27 points
//     (the nine-lanes-per-point chain, then a = M^T r for its points), fetches the element's vertices and u itself and parks them in
//     a copy of its own, so the two never have to meet.
// One barrier per element: behind it the next element's operands are complete and the current element's have been read.  64 KB of LDS (two operand
// arrays, two coefficient tables, the chain's scratch no longer inside the operands), two workgroups per CU.
// Same arithmetic, same order of operations per value as hex27_blocks.hpp: the two kernels produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include "assemble_kernels.hpp"
#include "device_common.hpp"
#include "hex27_blocks.hpp"

namespace fenris_hip {

struct Hex27RolesLds {
    static constexpr int N = 27, NG = 8, NQ = 27, RP = 28, QS = 29, AW = 3 * RP * QS;
    static constexpr int o_A = 0;                       // [2][c][RP][QS]
    static constexpr int o_coef = 2 * AW;               // [2][9][28]
    static constexpr int o_M = o_coef + 2 * 9 * 28;     // [28][9]
    static constexpr int o_sJ = o_M + 28 * 9;           // chain scratch (hex27_blocks.hpp keeps it inside A): [28][9] each, then [27][27]
    static constexpr int o_sI = o_sJ + 28 * 9;
    static constexpr int o_sH = o_sI + 28 * 9;
    static constexpr int o_sP = o_sH + 28 * 9;
    static constexpr int o_ggeom = o_sP + 27 * 27 + 3;
    static constexpr int o_qw = o_ggeom + NQ * NG * 3;
    static constexpr int NPW = 4;                       // prologue wavefronts: seven points each (six for the last)
    static constexpr int o_X = o_qw + 28;               // [NPW][24]
    static constexpr int o_U = o_X + NPW * 24;          // [NPW][82]
    static constexpr int total = o_U + NPW * 82;
    static constexpr int KE_TRI = Hex27BlkLds::KE_TRI;
};

template <int OP>
__global__ void __launch_bounds__(512, 4) k_hex27_dense_roles(const KArgs a, double mu_u, double lambda_u) {
    using L = Hex27RolesLds;
    constexpr int N = L::N, NG = L::NG, NQ = L::NQ, RP = L::RP, QS = L::QS;
    constexpr bool NH = (OP == FH_NEO_HOOKEAN);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, nt = 512;
    for (int i = tid; i < NQ * NG * 3; i += nt) lds[L::o_ggeom + i] = a.ggeom[i];
    for (int i = tid; i < 28; i += nt) lds[L::o_qw + i] = (i < NQ) ? a.qw[i] : 0.0;
    for (int i = tid; i < L::o_ggeom; i += nt) lds[i] = 0.0;   // both operand arrays (with their rows / points of zeros), coefficients, M, scratch
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long Gs = gridDim.x;
    const long long w0 = a.work_begin + blockIdx.x;
    if (w0 >= a.work_end) return;
    const int n_el = (int)((a.work_end - w0 + Gs - 1) / Gs);   // elements of this workgroup: w0, w0 + Gs, ...
    auto elem_of = [&](long long w) { const long long wc = min(w, a.work_end - 1); return a.labels ? (long long)a.labels[wc] : wc; };

    if (wave >= 4) {
        // ================================================================================================ prologue wavefronts
        const int pw = wave - 4, lane = tid & 63;
        __builtin_amdgcn_s_setprio(2);
        double* const Xs = lds + L::o_X + 24 * pw;
        double* const Us = lds + L::o_U + 82 * pw;
        // the element's 24 + 81 inputs, two per lane (items lane and lane + 64), as a chain of three requests like in hex27_blocks.hpp
        auto item_node = [&](int ri, long long e) { return a.conn[(size_t)e * N + (ri < 24 ? ri / 3 : (ri - 24) / 3)]; };
        auto item_value = [&](int ri, int node) {
            const bool isu = ri >= 24;
            const double* src = (isu && a.u) ? a.u : a.verts;
            return src[(size_t)node * 3 + (isu ? (ri - 24) % 3 : ri % 3)];
        };
        const int ri0 = lane, ri1 = min(lane + 64, 104);
        long long e_cur = elem_of(w0), e_n1 = elem_of(w0 + Gs), e_n2 = elem_of(w0 + 2 * Gs);
        double v0_cur = item_value(ri0, item_node(ri0, e_cur)), v1_cur = item_value(ri1, item_node(ri1, e_cur));
        int n0_n1 = item_node(ri0, e_n1), n1_n1 = item_node(ri1, e_n1);
        // chain lanes: nine per point; this wavefront's points are [7 pw, min(7 pw + 7, 27))
        const int pl = lane / 9, pe = lane - 9 * pl, pi = pe / 3, pj = pe - 3 * pi;
        const int q_lo = 7 * pw, q_hi = min(q_lo + 7, NQ);
        const auto gref_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.gref), (short)0, NQ * N * 3 * 8, 0x00020000);
        const auto greft_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(NH ? a.gref_t : a.gref), (short)0, NQ * N * 3 * 8, 0x00020000);
        typedef unsigned ld_u32x2 __attribute__((ext_vector_type(2)));
        const int n_items = (q_hi - q_lo) * N;   // (point, node) pairs of phase P2: 189 or 162
        // Everything of the prologue that does not depend on the element lives in registers for the whole sweep -- these wavefronts have the room the
        // alternating form does not: the lane's nine reference gradients of the chain, its nine of phase P2, the point's parameters.
        const int q_raw = q_lo + pl;
        const bool pvalid = pl < 7 && q_raw < q_hi;
        const int pq = min(q_raw, q_hi - 1);
        double r9[9], g3[3][3];
        if (NH) {
            const unsigned vt = (unsigned)((pi * NQ + pq) * 24 + pj * 8);
#pragma unroll
            for (int j = 0; j < 9; ++j)
                r9[j] = __builtin_bit_cast(double, (ld_u32x2)__builtin_amdgcn_raw_buffer_load_b64(greft_rsrc, vt, j * 3 * NQ * 24, 0));
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
                g3[k][cc] = __builtin_bit_cast(double, (ld_u32x2)__builtin_amdgcn_raw_buffer_load_b64(gref_rsrc, (unsigned)((q_lo * N + lane + 64 * k) * 24 + cc * 8), 0, 0));
        const double mu = a.qparams ? a.qparams[2 * pq] : mu_u, lambda = a.qparams ? a.qparams[2 * pq + 1] : lambda_u;
        for (int it = 0; it < n_el; ++it) {
            const long long e = e_cur;
            double* const A = lds + L::o_A + (it & 1) * L::AW;
            double* const coef = lds + L::o_coef + (it & 1) * (9 * 28);
            // ---- P0: inputs of this element to the wavefront's own copy; requests for the next elements
            if (ri0 < 24) Xs[ri0] = v0_cur; else Us[ri0 - 24] = a.u ? v0_cur : 0.0;
            if (lane + 64 < 105) Us[ri1 - 24] = a.u ? v1_cur : 0.0;
            const double v0_n1 = item_value(ri0, n0_n1), v1_n1 = item_value(ri1, n1_n1);
            const int n0_n2 = item_node(ri0, e_n2), n1_n2 = item_node(ri1, e_n2);
            const long long e_n3 = elem_of(w0 + (long long)(it + 3) * Gs);
            asm volatile("" ::: "memory");
            // ---- P1: the chain of this wavefront's seven points   (FENRIS_HIP_ABLATE, timing only: 1 no chain / P2, 2 no matrix instructions, 4 no stores)
            if (!(a.ablate & 1)) {
                double* sJ = lds + L::o_sJ + pq * 9;
                double* sI = lds + L::o_sI + pq * 9;
                double* sH = lds + L::o_sH + pq * 9;
                double* sP = lds + L::o_sP + pq * 27;
                double Jv = 0.0;
#pragma unroll
                for (int g = 0; g < NG; ++g) Jv = fma(Xs[g * 3 + pi], lds[L::o_ggeom + (pq * NG + g) * 3 + pj], Jv);
                if (pvalid) sJ[pe] = Jv;
                if (NH) {
                    double P0 = 0.0, P1 = 0.0, P2 = 0.0;
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const double* un = Us + (3 * j + pi) * 3;
                        P0 = fma(un[0], r9[j], P0);
                        P1 = fma(un[1], r9[j], P1);
                        P2 = fma(un[2], r9[j], P2);
                    }
                    if (pvalid) { sP[pi * 9 + 0 + pj] = P0; sP[pi * 9 + 3 + pj] = P1; sP[pi * 9 + 6 + pj] = P2; }
                }
                asm volatile("" ::: "memory");
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
                const double s = lds[L::o_qw + pq] * fabs(detJ);
                double Mv = Jiv, Bv = (pi == pj) ? 1.0 : 0.0, c_a = -(s * mu);
                const double c_l = s * lambda, c_m = s * mu;
                if (NH) {
                    const double Hv = (sP[0 + pe] + sP[9 + pe]) + sP[18 + pe];
                    asm volatile("" ::: "memory");
                    if (pvalid) { sI[pe] = Jiv; sH[pe] = Hv; }
                    asm volatile("" ::: "memory");
                    const double Fv = ((pi == pj) ? 1.0 : 0.0) + fma(sH[pi * 3 + 0], sI[0 + pj], fma(sH[pi * 3 + 1], sI[3 + pj], sH[pi * 3 + 2] * sI[6 + pj]));
                    asm volatile("" ::: "memory");
                    if (pvalid) sJ[pe] = Fv;
                    asm volatile("" ::: "memory");
                    double Jd;
                    {
                        const double F[3][3] = {{sJ[0], sJ[1], sJ[2]}, {sJ[3], sJ[4], sJ[5]}, {sJ[6], sJ[7], sJ[8]}};
                        Jd = det_small<3>(F);
                    }
                    double Fiv = 0.0;
                    if (Jd > 0.0) Fiv = (sJ[r1 * 3 + c1] * sJ[r2 * 3 + c2] - sJ[r2 * 3 + c1] * sJ[r1 * 3 + c2]) * (1.0 / Jd);
                    Bv = fma(sJ[pi * 3 + 0], sJ[pj * 3 + 0], fma(sJ[pi * 3 + 1], sJ[pj * 3 + 1], sJ[pi * 3 + 2] * sJ[pj * 3 + 2]));
                    asm volatile("" ::: "memory");
                    if (pvalid) sH[pe] = Fiv;
                    asm volatile("" ::: "memory");
                    Mv = fma(sI[pi * 3 + 0], sH[0 + pj], fma(sI[pi * 3 + 1], sH[3 + pj], sI[pi * 3 + 2] * sH[6 + pj]));
                    if (pe == 0) c_a = (Jd > 0.0) ? s * (-mu + lambda * hex27b_log(Jd)) : __builtin_nan("");
                    if (!(Jd > 0.0)) Bv = __builtin_nan("");
                }
                if (pvalid) {
                    lds[L::o_M + pq * 9 + pe] = Mv;
                    if (pi <= pj) coef[(3 + pe - (pi * (pi + 1)) / 2) * 28 + pq] = c_m * Bv;
                    if (pe == 0) {
                        const bool bad = NH && c_a != c_a;
                        const double cl = bad ? c_a : c_l;
                        coef[0 * 28 + pq] = cl;
                        coef[1 * 28 + pq] = -c_a;
                        coef[2 * 28 + pq] = cl - c_a;
                    }
                }
                asm volatile("" ::: "memory");
            }
            // ---- P2: a_n = M^T r_n for this wavefront's points (their M lies in LDS: written by this wavefront, in order)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int li = lane + 64 * k;
                if (li < n_items && !(a.ablate & 1)) {
                    const int gi = q_lo * N + li, q = gi / N, n = gi - q * N;
                    const double* Mq = lds + L::o_M + q * 9;
#pragma unroll
                    for (int c = 0; c < 3; ++c) A[(c * RP + n) * QS + q] = fma(Mq[0 * 3 + c], g3[k][0], fma(Mq[1 * 3 + c], g3[k][1], Mq[2 * 3 + c] * g3[k][2]));
                }
            }
            // the requests of this element are consumed: the next element's inputs
            v0_cur = v0_n1; v1_cur = v1_n1; n0_n1 = n0_n2; n1_n1 = n1_n2;
            e_cur = e_n1; e_n1 = e_n2; e_n2 = e_n3;
            lds_barrier();   // operands of element `it` complete (and the matrix wavefronts are done with the other half)
        }
        lds_barrier();       // (the matrix wavefronts' last one)
        return;
    }

    // ==================================================================================================== matrix wavefronts
    lds_barrier();   // element 0 is ready
    long long e_next = elem_of(w0);
    for (int it = 0; it < n_el; ++it) {
        const long long e = e_next;
        e_next = elem_of(w0 + (long long)(it + 1) * Gs);
        const double* A = lds + L::o_A + (it & 1) * L::AW;
        const double* coefp = lds + L::o_coef + (it & 1) * (9 * 28);
        const int role = (wave + (it & 3)) & 3;
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
        int lane_o = tid & 63;
        asm volatile("" : "+v"(lane_o));
        const int gq = (lane_o >> 2) & 3, idx = lane_o & 3, kq4 = lane_o >> 4;
        auto node_off = [&](unsigned tbl, int u) {
            const int B = (int)((tbl >> (3 * (gq + 4 * u))) & 7u);
            return (unsigned)((min(4 * B + idx, N) * QS + kq4) * 8);
        };
        const char* Ab = reinterpret_cast<const char*>(A);
        const char* Cb = reinterpret_cast<const char*>(coefp);
        auto ldA = [&](int comp, unsigned off) { return *reinterpret_cast<const double*>(Ab + off + (size_t)(comp * RP * QS * 8)); };
        auto ldC = [&](int k, unsigned off) { return *reinterpret_cast<const double*>(Cb + off + (size_t)(k * 28 * 8)); };
        auto mm = [](double x, double y, double acc) { return __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc, 0, 0, 0); };
        const unsigned long long ke_addr = reinterpret_cast<unsigned long long>(a.ke_out + (size_t)e * L::KE_TRI);
        const unsigned long long ke_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ke_addr >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)ke_addr);
        const auto ke_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ke_u), (short)0, L::KE_TRI * 8, 0x00020000);
        struct Unit { double k[3][3], m; };
        auto zero_unit = [](Unit& x) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) x.k[i][j] = 0.0;
            x.m = 0.0;
        };
        auto step_unit = [&](Unit& x, const double (&r)[3], const double (&wb)[3], const double (&cf)[9]) {
            const double s0 = cf[2] * r[0], s1 = cf[2] * r[1], s2 = cf[2] * r[2];
            const double h0 = fma(cf[5], r[2], fma(cf[4], r[1], cf[3] * r[0]));
            const double h1 = fma(cf[7], r[2], fma(cf[6], r[1], cf[4] * r[0]));
            const double h2 = fma(cf[8], r[2], fma(cf[7], r[1], cf[5] * r[0]));
            const double l0 = cf[0] * r[0], l1 = cf[0] * r[1], l2 = cf[0] * r[2];
            const double n0 = cf[1] * r[0], n1 = cf[1] * r[1], n2 = cf[1] * r[2];
            x.k[0][0] = mm(s0, wb[0], x.k[0][0]); x.k[1][1] = mm(s1, wb[1], x.k[1][1]); x.k[2][2] = mm(s2, wb[2], x.k[2][2]);
            x.m = mm(h0, wb[0], x.m);
            x.k[0][1] = mm(l0, wb[1], x.k[0][1]); x.k[0][2] = mm(l0, wb[2], x.k[0][2]); x.k[1][0] = mm(l1, wb[0], x.k[1][0]);
            x.k[1][2] = mm(l1, wb[2], x.k[1][2]); x.k[2][0] = mm(l2, wb[0], x.k[2][0]); x.k[2][1] = mm(l2, wb[1], x.k[2][1]);
            x.m = mm(h1, wb[1], x.m);
            x.k[0][1] = mm(n1, wb[0], x.k[0][1]); x.k[0][2] = mm(n2, wb[0], x.k[0][2]); x.k[1][0] = mm(n0, wb[1], x.k[1][0]);
            x.k[1][2] = mm(n2, wb[1], x.k[1][2]); x.k[2][0] = mm(n0, wb[2], x.k[2][0]); x.k[2][1] = mm(n1, wb[2], x.k[2][1]);
            x.m = mm(h2, wb[2], x.m);
        };
        auto store_unit = [&](Unit& x, int u, int ls) {
            typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
            typedef unsigned st_u32x2 __attribute__((ext_vector_type(2)));
            typedef double st_f64x2 __attribute__((ext_vector_type(2)));
            const int g = (ls >> 2) & 3;
            const int IB = (int)((ibt >> (3 * (g + 4 * u))) & 7u), JB = (int)((jbt >> (3 * (g + 4 * u))) & 7u);
            const int I = 4 * IB + (ls >> 4), J = 4 * JB + (ls & 3);
            const bool valid = I < N && J < N && I <= J && !(a.ablate & 4);
            const unsigned vo = valid ? (unsigned)((I * (53 - I)) / 2 + J) * 72u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < 3; ++i) x.k[i][i] += x.m;
            if (I == J) { x.k[1][0] = x.k[0][1]; x.k[2][0] = x.k[0][2]; x.k[2][1] = x.k[1][2]; }
            const double* f = &x.k[0][0];
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
#pragma unroll 1
            for (int ks = 0; ks < ((a.ablate & 2) ? 0 : 7); ++ks) {
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
            lds_barrier();   // this element's operands have been read (the accumulators hold everything): the prologue may overwrite them
            int ls = tid & 63;
            asm volatile("" : "+v"(ls));
            store_unit(x0, 0, ls);
            store_unit(x1, 1, ls);
        } else {
            unsigned oR0 = node_off(ibt, 0), oC0 = node_off(jbt, 0);
            Unit x0;
            zero_unit(x0);
#pragma unroll 1
            for (int ks = 0; ks < ((a.ablate & 2) ? 0 : 7); ++ks) {
                double r0[3], c0[3], cf[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) cf[k] = ldC(k, oC);
#pragma unroll
                for (int c = 0; c < 3; ++c) { r0[c] = ldA(c, oR0); c0[c] = ldA(c, oC0); }
                step_unit(x0, r0, c0, cf);
                oC += 32u; oR0 += 32u; oC0 += 32u;
            }
            lds_barrier();
            int ls = tid & 63;
            asm volatile("" : "+v"(ls));
            store_unit(x0, 0, ls);
        }
    }
}

}  // namespace fenris_hip
