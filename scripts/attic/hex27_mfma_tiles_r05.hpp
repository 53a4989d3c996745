// Dense element matrices of tri-quadratic hexahedra (Hex27, s = 3) with the fp64 matrix cores: first pass of the
// two-pass owner-computes assembly for LinearElastic and NeoHookean operators with a uniform quadrature table.
//
// Per quadrature point the stress contraction of both materials has the form (materials.rs:108-118, 302-313)
//     C(I, J)[i][j] = c_l a_I[i] a_J[j]  -  c_a a_J[i] a_I[j]  +  delta_ij c_m g_I . g_J
// with g_n the physical gradients, a_n = F^-T g_n (NeoHookean) or a_n = g_n (LinearElastic) and
//     NeoHookean:     c_l = s lambda,  c_a = s alpha = s (-mu + lambda ln det F),  c_m = s mu      (s = w |det J|)
//     LinearElastic:  c_l = s lambda,  c_a = -s mu,                                c_m = s mu.
// For a component pair (i, j) the 27 x 27 matrix over the node pairs is therefore a sum of Gram-type products
//     K_ij = (c_l A_i) A_j^T - (c_a A_j) A_i^T  [+ delta_ij sum_k (c_m G_k) G_k^T],     A_k, G_k: 27 nodes x 27 points
// = "B^T D B" with K = 54 (+81): exactly the shape v_mfma_f64_16x16x4_f64 wants.  One workgroup (4 wavefronts) per
// element: a cooperative prologue leaves G_k, A_k (rows padded to 32, points to 28, zeros) and the coefficients in
// LDS; wavefront w owns the 16 x 16 tile (w >> 1, w & 1) of the six K_ij with i <= j and of the trace term (84 MFMAs; the
// components below the diagonal are transposed copies, see the loop); the fragments are stored straight to the PLANAR
// dense layout ke[e][I][i][j][J] (J fastest: 16 lanes = 128 contiguous bytes; node-major since round 4: the 9 x 27 doubles that the
// second pass, k_rows_from_dense, reads for one (element, local node) entry are ONE run of 1944 bytes instead of nine runs of 216
// that each straddle two or three cache lines).  fp64 MFMA peaks at the vector fp64 rate on CDNA4, so
// the gain is not flops but operand traffic: 2 LDS doubles per lane feed 1024 FMAs (0.002 reads per FMA per lane
// against 0.5 in the VALU pair loop).  Measured (C4, 200 k elements): VALU element kernel 9.85 ms, this kernel 6.0 ms, of
// which the 588 MFMAs per element take 2.9 ms (the fp64 MFMA floor is 3.06 ms) and do not overlap with the prologue
// of the second resident workgroup -- starting the two workgroups of a CU half a period apart changed nothing, i.e.
// fp64 MFMA and fp64 VALU work do not run side by side on gfx950.  FENRIS_HIP_ABLATE bits 1/2/4 skip prologue / MFMA /
// stores for such measurements (instrumented instantiation only, i.e. together with FENRIS_HIP_TRACE=1: in the production kernel a store
// under a run-time switch is a store the compiler cannot count on, see the stores).
#pragma once
#include <hip/hip_runtime.h>

#include "assemble_kernels.hpp"
#include "device_common.hpp"

namespace fenris_hip {

typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));
#ifndef HEX27_PREFETCH_OPS_VALUE
#define HEX27_PREFETCH_OPS_VALUE 0
#endif
#ifndef HEX27_PRIO_VALUE
#define HEX27_PRIO_VALUE 1
#endif
constexpr bool HEX27_PRIO = HEX27_PRIO_VALUE != 0;
constexpr bool HEX27_PREFETCH_OPS = HEX27_PREFETCH_OPS_VALUE != 0;   // (unused since the two-round form)

// log() out of line: inlined, the compiler keeps the coefficients of its polynomial in registers ACROSS the element loop (ten of them, hoisted as
// loop invariants) -- with three workgroups per CU that is what spilled, and a spill is a scratch access with a full vmcnt wait in phase P5
static __device__ __attribute__((noinline)) double hex27_log(double x) { return log(x); }

struct Hex27Lds {
    // RP rows per component: the 27 nodes and ONE row of zeros -- the matrix-core tiles are 16 x 16, rows 27 .. 31 of the second tile all read
    // row 27 (round 5; they used to have five rows of zeros of their own).  QS: point stride (odd: bank spread).
    // Round 5: the reference gradients of the basis (27 x 27 x 3 doubles, 17.5 KB) are no longer staged here -- every lane needs the same
    // nine of them for every element and fetches them (L1 / L2-resident) ahead of the previous element's stores; with the 5.5 KB of padding
    // rows gone the workgroup takes 52.9 instead of 75.9 KB, which leaves room on the CU for the row gather of the previous chunk
    // (engine_two_pass.hip, the overlapped form).
    static constexpr int N = 27, NG = 8, NQ = 27, RP = 28, QS = 29;
    static constexpr int o_ggeom = 0;                      // [q][g][3]
    static constexpr int o_qw = o_ggeom + NQ * NG * 3;     // [q]
    static constexpr int o_X = o_qw + 28;                  // [g][3]
    static constexpr int o_U = o_X + NG * 3;               // [n][3]
    static constexpr int o_Jinv = o_U + N * 3;             // [q][9]  J^-1, row-major
    static constexpr int o_s = o_Jinv + NQ * 9 + 1;        // [q]     w |det J|
    static constexpr int o_dF = o_s + 28;                  // [q]     det F
    static constexpr int o_gu = o_dF + 28;                 // [q][k][c] grad u (d x s); before that: J of the point (P1)
    static constexpr int o_Fi = o_gu + NQ * 9 + 1;         // [q][9]  F^-1
    static constexpr int o_coef = o_Fi + NQ * 9 + 1;       // [6][28] c_l, -c_a, sign sqrt|c_l - c_a|, sqrt|c_l - c_a|, sign sqrt|c_m|,
                                                           //         sqrt|c_m| (entry 27 = 0)
    static constexpr int o_G = o_coef + 6 * 28;            // [k][RP][QS]
    static constexpr int o_A = o_G + 3 * RP * QS;          // [k][RP][QS]
    static constexpr int total = o_A + 3 * RP * QS;
};

// FORM 0 (the default): 16 x 16 tiles, v_mfma_f64_16x16x4.  FORM 1 (round 5 experiment, FENRIS_HIP_HEX27_BLOCKS=1): the products as 4 x 4 x 4
// blocks, v_mfma_f64_4x4x4_4b -- see "matrix cores, second form" below: the instruction is 1.6 - 2 x faster per flop on this part and the form
// halves the matrix-core time, but the pass as a whole does not get faster (7.30 - 7.38 against 7.04 - 7.07 ms: with three workgroups per CU the
// latency of the prologue's seven phases is what remains; profiles/r05_c4_mfma_blocks.txt).
template <int OP, bool TRACE = false, int FORM = 0>
__global__ void __launch_bounds__(256, 3) k_hex27_dense_mfma(const KArgs a, double mu_u, double lambda_u) {
    using L = Hex27Lds;
    constexpr int N = L::N, NG = L::NG, NQ = L::NQ, RP = L::RP, QS = L::QS;
    constexpr bool NH = (OP == FH_NEO_HOOKEAN);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, nt = 256, lane = tid & 63, wave = tid >> 6;
    // tables, and zeros in the padding of G / A / coef (written once: the prologue only touches n < 27, q < 27)
    for (int i = tid; i < NQ * NG * 3; i += nt) lds[L::o_ggeom + i] = a.ggeom[i];
    for (int i = tid; i < 28; i += nt) lds[L::o_qw + i] = (i < NQ) ? a.qw[i] : 0.0;
    for (int i = tid; i < 2 * 3 * RP * QS; i += nt) lds[L::o_G + i] = 0.0;
    for (int i = tid; i < 6 * 28; i += nt) lds[L::o_coef + i] = 0.0;
    __syncthreads();
    double* G = lds + L::o_G;
    double* A = NH ? lds + L::o_A : G;  // LinearElastic: a_n = g_n
    // FENRIS_HIP_TRACE: cycles of wavefront 0 per phase (P0, P1, P2, P3, P4, P5, MFMA, stores, transposed stores), summed over
    // workgroups into trace[16 + phase]; trace[31] counts elements
    unsigned long long ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = 0, ph_n = 0;
    const bool tracing = TRACE && a.trace != nullptr;   // instrumented instantiation only: the counters cost registers
    auto mark = [&](int k) {
        if (tracing) { const unsigned long long t = __builtin_readcyclecounter(); ph[k] += t - ph_t; ph_t = t; }
    };

    // Coefficients of a point as the matrix-core loop wants them (one lane per point computes the roots once; every lane of
    // every k-step used to).  On the diagonal components the two terms share their operands, (c_l - c_a) a_I[i] a_J[i], and
    // the coefficient is split as sign * sqrt|c| * sqrt|c| over the two operands; NaN coefficients (det F <= 0) stay NaN.
    auto put_coef = [&](int q, double c_l, double c_a, double c_m) {
        // roots by v_rsq_f64 + Newton (a few ulp are as good as exact here: the two factors only have to multiply back to c to
        // rounding, and both mirror entries use the same pair); 0 and NaN map to themselves
        auto root = [](double x) {
            if (!(x > 0.0)) return x;
            const double y = rsqrt_newton(x), r = x * y;
            return fma(fma(-r, r, x), 0.5 * y, r);
        };
        lds[L::o_coef + q] = c_l;
        lds[L::o_coef + 28 + q] = -c_a;
        if constexpr (FORM == 1) {   // the coefficient goes to ONE operand (the mirror images are copies there, see the stores): no roots
            lds[L::o_coef + 56 + q] = c_l - c_a;
            lds[L::o_coef + 112 + q] = c_m;
            return;
        }
        const double cd = c_l - c_a, rd = root(fabs(cd)), rm = root(fabs(c_m));
        lds[L::o_coef + 56 + q] = copysign(rd, cd);
        lds[L::o_coef + 84 + q] = rd;
        lds[L::o_coef + 112 + q] = copysign(rm, c_m);
        lds[L::o_coef + 140 + q] = rm;
    };

    // The element's inputs -- 8 geometry vertices, 27 values of u: element index, node index, then the gather -- are prefetched
    // in registers as a chain of three requests, each consumed one element after it was issued: while element w is worked on,
    // the values of w + G, the node indices of w + 2 G and the element index of w + 3 G are on their way.  EVERY thread takes
    // part (threads without a role fetch a harmless duplicate) and nothing is fetched under a branch, so no wait stands
    // behind a request in the element that issues it (a divergent form of this cost a full memory latency per element).
    const bool xrole = tid < NG * 3, urole = NH && tid >= 64 && tid < 64 + N * 3;
    const int ri = xrole ? tid : (urole ? tid - 64 : 0);         // component index of this thread's value
    const double* src = (urole && a.u) ? a.u : a.verts;           // u = NULL: zeros
    const bool val_used = xrole || (urole && a.u);
    const long long Gs = gridDim.x;
    auto elem_of = [&](long long w) { const long long wc = min(w, a.work_end - 1); return a.labels ? (long long)a.labels[wc] : wc; };
    auto node_at = [&](long long e) { return a.conn[(size_t)e * N + ri / 3]; };
    auto value_at = [&](int node) { return src[(size_t)node * 3 + ri % 3]; };
    const long long w0 = a.work_begin + blockIdx.x;
    if (w0 >= a.work_end) return;
    long long e_cur = elem_of(w0), e_n1 = elem_of(w0 + Gs), e_n2 = elem_of(w0 + 2 * Gs);
    double val_cur = value_at(node_at(e_cur));
    int node_n1 = node_at(e_n1);
    // reference gradients of this thread's three (point, node) items of phase P2 (item = tid + 256 k): the same for every element, but
    // fetched anew for each -- ahead of the previous element's stores, see there -- so that they are not live across the matrix-core loop
    double gr[3][3];
    auto load_gref = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int it = min(tid + nt * k, NQ * N - 1);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) gr[k][cc] = a.gref[it * 3 + cc];
        }
    };
    load_gref();
    // nothing pending at the loop's entry: otherwise the compiler takes the distance from these first requests to their use
    // (a handful of operations) for every trip, and each element starts by waiting for the previous element's stores
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    for (long long w = w0; w < a.work_end; w += Gs) {
        const long long e = e_cur;
        if (tracing) { ph_t = __builtin_readcyclecounter(); ++ph_n; }
        // P0: geometry nodes (the first 8) and u of the element, from the registers; requests for the next elements
        if (xrole) lds[L::o_X + ri] = val_cur;
        if (urole) lds[L::o_U + ri] = val_used ? val_cur : 0.0;
        double val_n1 = value_at(node_n1);
        int node_n2 = node_at(e_n2);
        long long e_n3 = elem_of(w + 3 * Gs);
        lds_barrier();
        mark(0);
        // The prologue's fp64 chains share the SIMD's fp64 datapath with the matrix instructions of the other workgroups of the CU (64 cycles
        // each, issued back to back by their wavefronts): at equal priority a dependent chain of ~160 vector instructions waits behind bursts
        // of them.  Raised priority for the prologue, normal priority for the matrix-core rounds: a vector instruction then waits for at most
        // the matrix instruction in flight.
        if (HEX27_PRIO) __builtin_amdgcn_s_setprio(3);
        // The 3 x 3 work of a point (Jacobian, its inverse, F, its inverse) is spread over nine lanes, one per entry: a wavefront
        // that runs it alone, one lane per point, issues ~250 dependent fp64 instructions while the other three wait at the
        // barrier -- and while the other workgroup of the CU multiplies, each of them waits for a 64-cycle matrix instruction.
        const int pq = min(tid / 9, NQ - 1), pi = (tid % 9) / 3, pj = tid % 3;     // (point, row, column) of this lane
        // entry (pi, pj) of adj(m) r for the matrix at m9 (row-major in LDS): cofactor (pj, pi), the expressions of adj_scaled
        auto inverse_entry = [&](const double* m9, double r) {
            const int r1 = (pj + 1) % 3, r2 = (pj + 2) % 3, c1 = (pi + 1) % 3, c2 = (pi + 2) % 3;
            return (m9[r1 * 3 + c1] * m9[r2 * 3 + c2] - m9[r2 * 3 + c1] * m9[r1 * 3 + c2]) * r;
        };
        // P1a: J = X G^T (hexahedron.rs:324-326 -> :101-107), one entry per lane, parked where grad u goes later
        if (tid < NQ * 9 && !(TRACE && (a.ablate & 1))) {
            double t = 0.0;
            for (int g = 0; g < NG; ++g) t = fma(lds[L::o_X + g * 3 + pi], lds[L::o_ggeom + (pq * NG + g) * 3 + pj], t);
            lds[L::o_gu + tid] = t;
        }
        lds_barrier();
        // P1b: inverse entry; the lane of entry (0, 0) also leaves s = w |det J| and, for LinearElastic, the coefficients
        if (tid < NQ * 9 && !(TRACE && (a.ablate & 1))) {
            const double* m9 = lds + L::o_gu + pq * 9;
            const double J[3][3] = {{m9[0], m9[1], m9[2]}, {m9[3], m9[4], m9[5]}, {m9[6], m9[7], m9[8]}};
            const double detJ = det_small<3>(J);
            double v = 0.0;
            if (detJ == 0.0) {  // try_inverse fails only for det == 0 exactly (elliptic.rs:401-404)
                if (pi == 0 && pj == 0) report_singular(a.status, e);
            } else {
                v = inverse_entry(m9, 1.0 / detJ);
            }
            lds[L::o_Jinv + tid] = v;
            if (pi == 0 && pj == 0) {
                const double s = lds[L::o_qw + pq] * fabs(detJ);  // elliptic.rs:422
                lds[L::o_s + pq] = s;
                if (!NH) {
                    const double mu = a.qparams ? a.qparams[2 * pq] : mu_u, lambda = a.qparams ? a.qparams[2 * pq + 1] : lambda_u;
                    put_coef(pq, s * lambda, -(s * mu), s * mu);
                }
            }
        }
        lds_barrier();
        mark(1);
        // P2: one lane per (point, node): g_n = J^-T grad_ref phi_n
        if (!(TRACE && (a.ablate & 1))) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int it = tid + nt * k;
                if (it < NQ * N) {
                    const int q = it / N, n = it % N;
                    const double* Ji = lds + L::o_Jinv + q * 9;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        G[(i * RP + n) * QS + q] = fma(Ji[0 * 3 + i], gr[k][0], fma(Ji[1 * 3 + i], gr[k][1], Ji[2 * 3 + i] * gr[k][2]));
                }
            }
        }
        lds_barrier();
        mark(2);
        if (NH && !(TRACE && (a.ablate & 1))) {
            // P3: grad u (d x s) = sum_n g_n u_n^T, one lane per (point, k, c)
            for (int it = tid; it < NQ * 9; it += nt) {
                const int q = it / 9, k = (it % 9) / 3, c = it % 3;
                double t = 0.0;
                for (int n = 0; n < N; ++n) t = fma(G[(k * RP + n) * QS + q], lds[L::o_U + n * 3 + c], t);
                lds[L::o_gu + it] = t;
            }
            lds_barrier();
            mark(3);
            // P4: F = I + (grad u)^T (fenris-solid/src/lib.rs:20-29), one entry of F^-1 per lane; det F goes to the lane that
            // computes the point's coefficients in the next phase (J <= 0 => NaN block there, zeros here)
            if (tid < NQ * 9) {
                const double* gq = lds + L::o_gu + pq * 9;
                double F9[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) F9[i * 3 + j] = (i == j ? 1.0 : 0.0) + gq[j * 3 + i];
                const double F[3][3] = {{F9[0], F9[1], F9[2]}, {F9[3], F9[4], F9[5]}, {F9[6], F9[7], F9[8]}};
                const double Jd = det_small<3>(F);
                double v = 0.0;
                if (Jd > 0.0) {
                    // the cofactor of this lane from F: entry (r, c) of F is delta + gu[c][r]
                    const int r1 = (pj + 1) % 3, r2 = (pj + 2) % 3, c1 = (pi + 1) % 3, c2 = (pi + 2) % 3;
                    auto Fe = [&](int r, int c) { return (r == c ? 1.0 : 0.0) + gq[c * 3 + r]; };
                    v = (Fe(r1, c1) * Fe(r2, c2) - Fe(r2, c1) * Fe(r1, c2)) * (1.0 / Jd);
                }
                lds[L::o_Fi + tid] = v;
                if (pi == 0 && pj == 0) lds[L::o_dF + pq] = Jd;
            }
            lds_barrier();
            mark(4);
            // P5: a_n = F^-T g_n on the first three wavefronts; meanwhile the fourth computes the coefficients of
            // materials.rs:287-315, one lane per point (a logarithm and two roots: as long as the other three's share)
            if (wave == 3) {
                if (lane < NQ) {
                    const int q = lane;
                    const double Jd = lds[L::o_dF + q], s = lds[L::o_s + q];
                    const double mu = a.qparams ? a.qparams[2 * q] : mu_u, lambda = a.qparams ? a.qparams[2 * q + 1] : lambda_u;
                    if (Jd <= 0.0) {
                        const double nan = __builtin_nan("");
                        put_coef(q, nan, nan, nan);
                    } else {
                        put_coef(q, s * lambda, s * (-mu + lambda * hex27_log(Jd)), s * mu);
                    }
                }
            } else {
                for (int it = tid; it < NQ * N; it += 192) {
                    const int q = it / N, n = it % N;
                    const double* Fi = lds + L::o_Fi + q * 9;
                    const double g0 = G[(0 * RP + n) * QS + q], g1 = G[(1 * RP + n) * QS + q], g2 = G[(2 * RP + n) * QS + q];
#pragma unroll
                    for (int i = 0; i < 3; ++i) A[(i * RP + n) * QS + q] = fma(Fi[0 * 3 + i], g0, fma(Fi[1 * 3 + i], g1, Fi[2 * 3 + i] * g2));
                }
            }
            lds_barrier();
            mark(5);
        }
        if constexpr (FORM == 1) {
        // ---- matrix cores, second form (round 5): 4 x 4 x 4 blocks.
        // Measured on this part (scripts/ubench/mfma_f64_rate.hip, profiles/r05_c4_mfma_blocks.txt): v_mfma_f64_16x16x4 issues once per ~100 cycles
        // and SIMD whatever the number of wavefronts and accumulators (46 - 48 TFLOP/s, 61 % of the 78.6 the part is specified with; 33 - 35 with
        // one wavefront per SIMD), v_mfma_f64_4x4x4_4b -- four independent 4 x 4 x 4 products per instruction -- reaches 75 TFLOP/s (55 - 68 with
        // one wavefront).  The 16 x 16 tiles below are bound by exactly that.  The block form pads 27 to 28 instead of 32 and leaves out the
        // mirrored blocks of the symmetric components row block by row block.
        //   * Register layout (found by experiment, scripts/ubench/mfma_f64_4x4_layout.hip): A: lane = (i + 4 g) + 16 k, B: lane = (j + 4 g) + 16 k,
        //     D: lane = (j + 4 g) + 16 i, g = the block.  Which node block a group g holds is the lane's choice of LDS address.
        //   * Third version of this form: ROW BROADCAST x COLUMN WINDOW.  The A operand holds ONE row block IB in all four groups (the four groups
        //     read the same addresses: a broadcast), the B operand a window of sixteen consecutive nodes (four consecutive column blocks: the
        //     conflict-free access the row stride of 29 was chosen for).  The result is then rows 4 IB .. 4 IB + 3 x sixteen consecutive columns:
        //     the direct stores write runs of sixteen doubles like the tiles do.  Windows: W0 = nodes 0 .. 15, W1 = 16 .. 27 (+ padding), and for
        //     row block 3 of the symmetric components Wx = 12 .. 27 (its four upper blocks in one instruction).
        //     (First version: seven rotated column arrangements against two row arrangements -- 9 fetches per 14 instructions but wrapped
        //     arrangements with bank conflicts, 32-byte store pieces and one wavefront with 3.4 x the fetches of the others: 7.80 ms against 7.37.)
        //   * Work: wavefront role r = 0, 1, 2 takes row blocks r and 6 - r (38 instructions per k-step), role 3 row block 3 and the component
        //     (1, 2) of row blocks 0 - 2 (30); roles rotate from element to element.  K_ii = its own part + the trace term, both in the registers
        //     of the wavefront that owns the row block.  144 instructions of 512 flop per k-step and workgroup (the tiles: 42 of 2 048).
        //   * The coefficient multiplies the A operand only; the lower blocks of the symmetric components and the lower halves of their diagonal
        //     blocks are stored as copies of the upper ones (bit-identical, like util.rs:38-51 mirrors): no square roots in the prologue.
        const int role = (__builtin_amdgcn_readfirstlane(wave) + (int)(((w - w0) / Gs) & 3)) & 3;   // rotates: every SIMD gets every role
        // (the lane id behind an empty asm: everything derived from it is formed per element instead of living in registers across the loop)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int l15 = lane_o & 15, idx = lane_o & 3, kq4 = lane_o >> 4;
        auto win_off = [&](int start) { return (unsigned)((min(start + l15, N) * QS + kq4) * 8); };      // window of 16 nodes from `start`
        auto row_off = [&](int IB) { return (unsigned)((min(4 * IB + idx, N) * QS + kq4) * 8); };        // row block IB in every group
        const char* Gb = reinterpret_cast<const char*>(G);
        const char* Ab = reinterpret_cast<const char*>(A);
        auto ldc = [&](const char* base, int comp, unsigned off) {   // `off` carries the k-step
            return *reinterpret_cast<const double*>(base + off + (size_t)(comp * RP * QS * 8));
        };
        auto mm = [](double x, double y, double acc) { return __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc, 0, 0, 0); };
        const unsigned long long ke_addr = reinterpret_cast<unsigned long long>(a.ke_out + (size_t)e * (81 * 81));
        const unsigned long long ke_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ke_addr >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)ke_addr);
        const auto ke_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ke_u), (short)0, 81 * 81 * 8, 0x00020000);
        auto put = [&](unsigned voff, int soff_doubles, double v) {
            typedef unsigned put_u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(put_u32x2, v), ke_rsrc, voff, soff_doubles * 8, 0);
        };
        constexpr unsigned DROP = 0x80000000u;   // an offset beyond the matrix: the buffer's bounds check drops the lane's store
        // offsets of the lane's entry (I, J) = (4 IB + i, start + lane & 15) for the stores of one (row block, window): direct and mirrored, for a
        // component off the diagonal (every entry inside the matrix) and for a symmetric one (direct: blocks on and above the diagonal, in the
        // diagonal block the entries on and above ITS diagonal; mirrored: the same without the diagonal itself)
        struct StOff { unsigned dir, mir, dirs, mirs; };
        auto st_off = [&](int IB, int start, int ls) {
            const int i = ls >> 4, J = start + (ls & 15), I = 4 * IB + i, JB = J >> 2, jj = J & 3;
            const bool valid = I < N && J < N;
            const bool upper = JB > IB || (JB == IB && i <= jj), strict = JB > IB || (JB == IB && i < jj);
            StOff o;
            o.dir = valid ? (unsigned)(I * (9 * N) + J) * 8u : DROP;
            o.mir = valid ? (unsigned)(J * (9 * N) + I) * 8u : DROP;
            o.dirs = (valid && upper) ? o.dir : DROP;
            o.mirs = (valid && strict) ? o.mir : DROP;
            return o;
        };
        auto rotate_inputs = [&]() {
            // The requests of this element are consumed here: not earlier (they need their time), and not behind the stores below
            asm volatile("" : "+v"(val_n1), "+v"(node_n2), "+v"(e_n3));
            val_cur = val_n1;
            node_n1 = node_n2;
            e_cur = e_n1;
            e_n1 = e_n2;
            e_n2 = e_n3;
        };
        if (HEX27_PRIO) __builtin_amdgcn_s_setprio(0);
        unsigned oC = (unsigned)(kq4 * 8);
        unsigned oW0 = win_off(0), oW1 = win_off(16);
        if (role < 3) {
            // ---- row blocks IBa = role (windows W0 and W1; components (0, 1), (0, 2) off the diagonal) and IBb = 6 - role (symmetric: W1 only)
            const int IBa = role, IBb = 6 - role;
            unsigned oRa = row_off(IBa), oRb = row_off(IBb);
            double aD[3][2], aM[2], a01[2], a02[2], bD[3], bM = 0.0, b01[2], b02[2], b12[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { aM[t] = 0.0; a01[t] = 0.0; a02[t] = 0.0; b01[t] = 0.0; b02[t] = 0.0; b12[t] = 0.0; aD[0][t] = 0.0; aD[1][t] = 0.0; aD[2][t] = 0.0; }
            bD[0] = 0.0; bD[1] = 0.0; bD[2] = 0.0;
            if (!(TRACE && (a.ablate & 2))) {
#pragma unroll 1
                for (int ks = 0; ks < 7; ++ks) {
                    double Wa[3][2], Wg[3][2], ra[3], rg[3], rb[3], rh[3];
                    const double cl = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds + L::o_coef) + oC);
                    const double nca = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds + L::o_coef + 28) + oC);
                    const double cd = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds + L::o_coef + 56) + oC);
                    const double cm = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds + L::o_coef + 112) + oC);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { ra[c] = ldc(Ab, c, oRa); rg[c] = ldc(Gb, c, oRa); }
#pragma unroll
                    for (int c = 0; c < 3; ++c) { Wa[c][0] = ldc(Ab, c, oW0); Wa[c][1] = ldc(Ab, c, oW1); Wg[c][0] = ldc(Gb, c, oW0); Wg[c][1] = ldc(Gb, c, oW1); }
#pragma unroll
                    for (int c = 0; c < 3; ++c) { rb[c] = ldc(Ab, c, oRb); rh[c] = ldc(Gb, c, oRb); }
                    {   // row block IBa
                        const double s0 = cd * ra[0], s1 = cd * ra[1], s2 = cd * ra[2], g0 = cm * rg[0], g1 = cm * rg[1], g2 = cm * rg[2];
                        const double l0 = cl * ra[0], n1 = nca * ra[1], n2 = nca * ra[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            aD[0][t] = mm(s0, Wa[0][t], aD[0][t]); aD[1][t] = mm(s1, Wa[1][t], aD[1][t]); aD[2][t] = mm(s2, Wa[2][t], aD[2][t]);
                            aM[t] = mm(g0, Wg[0][t], aM[t]); a01[t] = mm(l0, Wa[1][t], a01[t]); a02[t] = mm(l0, Wa[2][t], a02[t]);
                            aM[t] = mm(g1, Wg[1][t], aM[t]); a01[t] = mm(n1, Wa[0][t], a01[t]); a02[t] = mm(n2, Wa[0][t], a02[t]);
                            aM[t] = mm(g2, Wg[2][t], aM[t]);
                        }
                    }
                    {   // row block IBb: its upper blocks lie in W1
                        const double s0 = cd * rb[0], s1 = cd * rb[1], s2 = cd * rb[2], g0 = cm * rh[0], g1 = cm * rh[1], g2 = cm * rh[2];
                        const double l0 = cl * rb[0], l1 = cl * rb[1], n1 = nca * rb[1], n2 = nca * rb[2];
                        bD[0] = mm(s0, Wa[0][1], bD[0]); bD[1] = mm(s1, Wa[1][1], bD[1]); bD[2] = mm(s2, Wa[2][1], bD[2]);
                        bM = mm(g0, Wg[0][1], bM);
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            b01[t] = mm(l0, Wa[1][t], b01[t]); b02[t] = mm(l0, Wa[2][t], b02[t]); b12[t] = mm(l1, Wa[2][t], b12[t]);
                            if (t == 0) bM = mm(g1, Wg[1][1], bM);
                            b01[t] = mm(n1, Wa[0][t], b01[t]); b02[t] = mm(n2, Wa[0][t], b02[t]); b12[t] = mm(n2, Wa[1][t], b12[t]);
                            if (t == 0) bM = mm(g2, Wg[2][1], bM);
                        }
                    }
                    oC += 32u; oW0 += 32u; oW1 += 32u; oRa += 32u; oRb += 32u;
                }
            }
            rotate_inputs();
            mark(6);
            load_gref();
            asm volatile("" ::: "memory");
            if (!(TRACE && (a.ablate & 4))) {
                int ls = lane;
                asm volatile("" : "+v"(ls));
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const StOff o = st_off(IBa, 16 * t, ls);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double v = aD[i][t] + aM[t];
                        put(o.dirs, (i * 3 + i) * N, v);
                        put(o.mirs, (i * 3 + i) * N, v);
                    }
                    put(o.dir, (0 * 3 + 1) * N, a01[t]); put(o.mir, (1 * 3 + 0) * N, a01[t]);
                    put(o.dir, (0 * 3 + 2) * N, a02[t]); put(o.mir, (2 * 3 + 0) * N, a02[t]);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const StOff o = st_off(IBb, 16 * t, ls);
                    if (t == 1) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const double v = bD[i] + bM;
                            put(o.dirs, (i * 3 + i) * N, v);
                            put(o.mirs, (i * 3 + i) * N, v);
                        }
                    }
                    put(o.dir, (0 * 3 + 1) * N, b01[t]); put(o.mir, (1 * 3 + 0) * N, b01[t]);
                    put(o.dir, (0 * 3 + 2) * N, b02[t]); put(o.mir, (2 * 3 + 0) * N, b02[t]);
                    put(o.dir, (1 * 3 + 2) * N, b12[t]); put(o.mir, (2 * 3 + 1) * N, b12[t]);
                }
            }
        } else {
            // ---- row block 3 (symmetric components against the window 12 .. 27) and the component (1, 2) of row blocks 0, 1, 2
            unsigned oWx = win_off(12), oR3 = row_off(3), oQ0 = row_off(0), oQ1 = row_off(1), oQ2 = row_off(2);
            double cD[3], cM = 0.0, c01[2], c02[2], c12[2], d12[3][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { c01[t] = 0.0; c02[t] = 0.0; c12[t] = 0.0; d12[0][t] = 0.0; d12[1][t] = 0.0; d12[2][t] = 0.0; }
            cD[0] = 0.0; cD[1] = 0.0; cD[2] = 0.0;
            if (!(TRACE && (a.ablate & 2))) {
#pragma unroll 1
                for (int ks = 0; ks < 7; ++ks) {
                    double Wa[3][2], Xa[3], Xg[3], r3[3], g3[3], q1[3], q2[3];
                    const double cl = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds + L::o_coef) + oC);
                    const double nca = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds + L::o_coef + 28) + oC);
                    const double cd = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds + L::o_coef + 56) + oC);
                    const double cm = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds + L::o_coef + 112) + oC);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { r3[c] = ldc(Ab, c, oR3); g3[c] = ldc(Gb, c, oR3); }
#pragma unroll
                    for (int c = 0; c < 3; ++c) { Xa[c] = ldc(Ab, c, oWx); Xg[c] = ldc(Gb, c, oWx); Wa[c][0] = ldc(Ab, c, oW0); Wa[c][1] = ldc(Ab, c, oW1); }
                    q1[0] = ldc(Ab, 1, oQ0); q1[1] = ldc(Ab, 1, oQ1); q1[2] = ldc(Ab, 1, oQ2);
                    q2[0] = ldc(Ab, 2, oQ0); q2[1] = ldc(Ab, 2, oQ1); q2[2] = ldc(Ab, 2, oQ2);
                    {
                        const double s0 = cd * r3[0], s1 = cd * r3[1], s2 = cd * r3[2], g0 = cm * g3[0], g1 = cm * g3[1], g2 = cm * g3[2];
                        const double l0 = cl * r3[0], l1 = cl * r3[1], n1 = nca * r3[1], n2 = nca * r3[2];
                        cD[0] = mm(s0, Xa[0], cD[0]); cD[1] = mm(s1, Xa[1], cD[1]); cD[2] = mm(s2, Xa[2], cD[2]);
                        cM = mm(g0, Xg[0], cM);
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            c01[t] = mm(l0, Wa[1][t], c01[t]); c02[t] = mm(l0, Wa[2][t], c02[t]); c12[t] = mm(l1, Wa[2][t], c12[t]);
                            if (t == 0) cM = mm(g1, Xg[1], cM);
                            c01[t] = mm(n1, Wa[0][t], c01[t]); c02[t] = mm(n2, Wa[0][t], c02[t]); c12[t] = mm(n2, Wa[1][t], c12[t]);
                            if (t == 0) cM = mm(g2, Xg[2], cM);
                        }
                    }
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const double l1 = cl * q1[b], n2 = nca * q2[b];
#pragma unroll
                        for (int t = 0; t < 2; ++t) d12[b][t] = mm(l1, Wa[2][t], d12[b][t]);
#pragma unroll
                        for (int t = 0; t < 2; ++t) d12[b][t] = mm(n2, Wa[1][t], d12[b][t]);
                    }
                    oC += 32u; oW0 += 32u; oW1 += 32u; oWx += 32u; oR3 += 32u; oQ0 += 32u; oQ1 += 32u; oQ2 += 32u;
                }
            }
            rotate_inputs();
            mark(6);
            load_gref();
            asm volatile("" ::: "memory");
            if (!(TRACE && (a.ablate & 4))) {
                int ls = lane;
                asm volatile("" : "+v"(ls));
                {
                    const StOff o = st_off(3, 12, ls);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double v = cD[i] + cM;
                        put(o.dirs, (i * 3 + i) * N, v);
                        put(o.mirs, (i * 3 + i) * N, v);
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const StOff o = st_off(3, 16 * t, ls);
                    put(o.dir, (0 * 3 + 1) * N, c01[t]); put(o.mir, (1 * 3 + 0) * N, c01[t]);
                    put(o.dir, (0 * 3 + 2) * N, c02[t]); put(o.mir, (2 * 3 + 0) * N, c02[t]);
                    put(o.dir, (1 * 3 + 2) * N, c12[t]); put(o.mir, (2 * 3 + 1) * N, c12[t]);
                }
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const StOff o = st_off(b, 16 * t, ls);
                        put(o.dir, (1 * 3 + 2) * N, d12[b][t]); put(o.mir, (2 * 3 + 1) * N, d12[b][t]);
                    }
            }
        }
        mark(7);
        lds_barrier();  // the next element's prologue overwrites G / A
        mark(8);
        continue;
        }
        // ---- matrix cores: wavefront w owns tile (tI, tJ) of the six K_ij with i <= j and of the trace term.
        // K_ji = K_ij^T (the element matrix is symmetric; util.rs:38-51 mirrors the upper triangle scalar by scalar), so the
        // three components below the diagonal are never multiplied: they are written as transposed copies -- 84 MFMAs per
        // wavefront instead of 147, and K_e symmetric bit for bit.  On the diagonal components the two terms share their
        // operands, (c_l - c_a) a_I[i] a_J[i], and the coefficient is split as sign * sqrt|c| * sqrt|c| over the two
        // operands: entry (I, J) and entry (J, I) are then sums of identical products in the same order, in whichever
        // tile they lie.
        // Which wavefront takes which tile rotates from element to element (round 5): tile (1, 0) of the three diagonal components and of the trace
        // term is the mirror image of tile (0, 1) and is no longer multiplied -- its wavefront skips round A, the wavefront of tile (0, 1) stores the
        // mirror image as well -- so one SIMD in four has half the matrix instructions of an element, and the rotation spreads that relief over
        // the four SIMDs (42 instead of 48 matrix instructions per k-step and workgroup).
        const int tile = (__builtin_amdgcn_readfirstlane(wave) + (int)(((w - w0) / Gs) & 3)) & 3;
        const bool light = tile == 2;
        const int tI = tile >> 1, tJ = tile & 1;
        const int rI = min(16 * tI + (lane & 15), N), rJ = min(16 * tJ + (lane & 15), N), kq = lane >> 4;   // rows 27 .. 31: the row of zeros
        // Two rounds over the seven k-steps (round 5: three workgroups per CU leave 168 registers; seven accumulator tiles at once were 56
        // of them): (A) the three diagonal components and the trace term, stored while (B) the three components above the diagonal multiply.
        // The operands are read from LDS twice (18 + 12 instead of 18 values per lane and k-step), the number of matrix instructions is the same.
        struct OpsA { double ar[3], ac[3], gr[3], gc[3], rds, rd, rms, rm; };
        struct OpsB { double ar[3], ac[3], cl, nca; };
        auto fetch_a = [&](int ks) {
            const int q = 4 * ks + kq;  // q = 27 is padding: operands and coefficients are zero there
            OpsA o;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o.ar[k] = A[(k * RP + rI) * QS + q];
                o.ac[k] = A[(k * RP + rJ) * QS + q];
                o.gr[k] = G[(k * RP + rI) * QS + q];
                o.gc[k] = G[(k * RP + rJ) * QS + q];
            }
            o.rds = lds[L::o_coef + 56 + q]; o.rd = lds[L::o_coef + 84 + q];
            o.rms = lds[L::o_coef + 112 + q]; o.rm = lds[L::o_coef + 140 + q];
            return o;
        };
        auto fetch_b = [&](int ks) {
            const int q = 4 * ks + kq;
            OpsB o;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o.ar[k] = A[(k * RP + rI) * QS + q];
                o.ac[k] = A[(k * RP + rJ) * QS + q];
            }
            o.cl = lds[L::o_coef + q]; o.nca = lds[L::o_coef + 28 + q];
            return o;
        };
        // store: C/D fragment of v_mfma_f64_16x16x4: col = lane & 15, row = (lane >> 4) + 4 reg
        // Buffer stores with the hardware's bounds check (round 5): the lanes of the padding rows / columns (I, J >= 27) get an offset
        // beyond the element's matrix and their store is dropped -- no branch, no EXEC masking.  Under `if (I < N && J < N)` every store sat
        // in a conditional block, and the compiler then counts NO store as certainly issued: the wait for the reference gradients
        // requested ahead of the stores became a wait for all the stores.
        const unsigned long long ke_addr = reinterpret_cast<unsigned long long>(a.ke_out + (size_t)e * (81 * 81));
        const unsigned long long ke_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ke_addr >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)ke_addr);
        const auto ke_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ke_u), (short)0, 81 * 81 * 8, 0x00020000);
        // address = base + per-lane offset (VGPR; beyond the matrix for a padding lane: only this part is bounds-checked) + a constant per
        // store (scalar offset): eight offset registers instead of one per store
        auto put = [&](unsigned voff, int soff_doubles, double v) {
            typedef unsigned put_u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(put_u32x2, v), ke_rsrc, voff, soff_doubles * 8, 0);
        };
        const int J = 16 * tJ + (lane & 15);
        const int I0 = 16 * tI + (lane >> 4);
        auto vo_row = [&](int reg) { return (I0 + 4 * reg < N && J < N) ? (unsigned)(I0 * (9 * N) + J) * 8u : 0x80000000u; };   // element (I0 + 4 reg, J) of a component
        auto vo_col = [&](int reg) { return (I0 + 4 * reg < N && J < N) ? (unsigned)(J * (9 * N) + I0) * 8u : 0x80000000u; };   // its mirror image (J, I0 + 4 reg)
        if (HEX27_PRIO) __builtin_amdgcn_s_setprio(0);
        // ---- round A: diagonal components and trace term
        {
            mfma_f64x4 accD[3], accM = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 3; ++i) accD[i] = mfma_f64x4{0, 0, 0, 0};
            if (!light && !(TRACE && (a.ablate & 2))) {
                // the next step's operands are fetched beside this one's products; two steps per trip with the two operand sets changing roles
                // (`cur = nxt` at the end of a one-step trip was sixteen register moves per step: a quarter of the kernel's vector instructions)
                auto mul_a = [&](const OpsA& o) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) accD[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.rds * o.ar[i], o.rd * o.ac[i], accD[i], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 3; ++k) accM = __builtin_amdgcn_mfma_f64_16x16x4f64(o.rms * o.gr[k], o.rm * o.gc[k], accM, 0, 0, 0);
                };
                // (fully unrolled: the two operand sets change roles without a copy; the compiler barrier keeps the fetches ONE step ahead -- all
                // seven steps' operands in flight at once are more registers than there are)
                OpsA oa[2];
                oa[0] = fetch_a(0);
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) {
                    if (ks < 6) oa[(ks + 1) & 1] = fetch_a(ks + 1);
                    mul_a(oa[ks & 1]);
                    asm volatile("" ::: "memory");
                }
            }
            // The requests of this element are consumed here: not earlier (they need their time), and not behind the stores below
            // (loads and stores share one in-order counter: waiting for a load issued before a store waits for the store).
            asm volatile("" : "+v"(val_n1), "+v"(node_n2), "+v"(e_n3));
            val_cur = val_n1;
            node_n1 = node_n2;
            e_cur = e_n1;
            e_n1 = e_n2;
            e_n2 = e_n3;
            if (!light && !(TRACE && (a.ablate & 4))) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const mfma_f64x4 v = accD[i] + accM;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) put(vo_row(reg), 4 * reg * (9 * N) + (i * 3 + i) * N, v[reg]);
                    if (tile == 1) {   // ... and tile (1, 0) of this component: the transpose
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) put(vo_col(reg), (i * 3 + i) * N + 4 * reg, v[reg]);
                    }
                }
            }
        }
        mark(6);
        // ---- round B: the components above the diagonal; K_ji = K_ij^T straight from the fragments: entry (I, J) of the tile goes to (J, I) of
        // the mirrored component.  Per store instruction a lane group writes four consecutive I of one J (32 bytes), and the four registers
        // complete the 128 bytes of that J; the L2 merges them.  (Staging the tile through LDS to store it row-wise cost a barrier, three LDS
        // round trips and the re-zeroing of the staging area: 2.4 k cycles per element against 1.x k.)
        {
            mfma_f64x4 accO[3];   // (0, 1), (0, 2), (1, 2)
#pragma unroll
            for (int i = 0; i < 3; ++i) accO[i] = mfma_f64x4{0, 0, 0, 0};
            if (!(TRACE && (a.ablate & 2))) {
                auto mul_b = [&](const OpsB& o) {
                    int t = 0;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = i + 1; j < 3; ++j, ++t) {
                            accO[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.cl * o.ar[i], o.ac[j], accO[t], 0, 0, 0);   // c_l a_I[i] a_J[j]
                            accO[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.nca * o.ar[j], o.ac[i], accO[t], 0, 0, 0);  // -c_a a_I[j] a_J[i]
                        }
                };
                OpsB ob[2];
                ob[0] = fetch_b(0);
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) {
                    if (ks < 6) ob[(ks + 1) & 1] = fetch_b(ks + 1);
                    mul_b(ob[ks & 1]);
                    asm volatile("" ::: "memory");
                }
            }
            mark(7);
            // The reference gradients of the NEXT element's phase P2 are requested here, ahead of the 24 stores of this round: P2 then waits for
            // them with those stores still in flight (issued behind the stores, the wait would drain them: a write latency per element; the
            // stores of round A are older, but a whole round older).  Not earlier: they would be live across this round's products.
            load_gref();
            asm volatile("" ::: "memory");
            if (!(TRACE && (a.ablate & 4))) {
                int t = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = i + 1; j < 3; ++j, ++t)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            put(vo_row(reg), 4 * reg * (9 * N) + (i * 3 + j) * N, accO[t][reg]);
                            put(vo_col(reg), (j * 3 + i) * N + 4 * reg, accO[t][reg]);
                        }
            }
        }
        lds_barrier();  // the next element's prologue overwrites G / A
        mark(8);
    }
    if (tracing && tid == 0) {
        for (int k = 0; k < 9; ++k) atomicAdd(a.trace + 16 + k, ph[k]);
        atomicAdd(a.trace + 31, ph_n);
    }
}

}  // namespace fenris_hip
