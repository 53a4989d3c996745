// Affine-element form of the owner-computes stiffness kernel (Hex8; Laplace / uniform LinearElastic).
//
// On an element whose trilinear map is affine (a parallelepiped) the Jacobian is constant, so the quadrature loop of
// elliptic.rs:398-432 collapses: with  Ghat_ab = sum_q w_q ghat_a(xi_q) ghat_b(xi_q)^T  (reference gradients only, built once
// per quadrature table on the host) the block of the node pair (a, b) is
//     G_ab = |det J| J^-T Ghat_ab J^-1,        K_ab = mu (tr G_ab I + G_ab^T) + lambda G_ab     (materials.rs:108-118 summed)
//                                              K_ab = tr G_ab                                   (laplace.rs:60-68)
// -- no per-point Jacobians, no physical gradients, no q-loop.  Which elements qualify is decided per element from the
// vertex coordinates (k_classify_affine_hex8); node blocks all of whose elements qualify run here, the others keep the
// general kernels (engine.hip: build_partition splits the sweep order by class).
//
// Work distribution: the row-owner lanes of rows_kernel.hpp (a lane owns an output block (owned node I, column J) and walks
// two terms (slot, local I, local J); blocks with more terms are split over 2 / 4 lanes and summed by DPP).  No LDS atomics;
// the finished rows are staged in LDS in CSR order and leave as 16-byte stores.  One barrier per node block: the staged
// rows, the per-slot Jacobian records and the parked table records are double-buffered by block parity.
//
// Exact symmetry (util.rs:38-51 mirrors the upper triangle): every term evaluates the canonical block (min(a, b), max(a, b))
// and transposes the result when a > b, the terms of a block are ordered by element id (k_build_row_lanes), and the split /
// DPP tree depends only on the number of terms -- so the owners of (I, J) and (J, I) add bitwise identical numbers in the
// same order.  Diagonal blocks mirror their upper triangle.  Run-to-run reproducible for the same reason.
#pragma once
#include <hip/hip_runtime.h>

#include "assemble_kernels.hpp"
#include "device_common.hpp"
#include "rows_kernel.hpp"

namespace fenris_hip {

struct AffineTables {
    const int* rec;      // [npos][rw]   GatherHdr (8 words) | slot list (us / 4 words, unused here) | row offsets (nbs + 1 words)
    const uint2* lanes;  // [npos][256]  lane records of rows_kernel.hpp
    const int* conn;     // [npos][cs]   geometry-node indices per slot, cs = 8 us
    const int* elem;     // [npos][us]   element id per slot (-1: empty slot)
    const double* ghat;  // [64][GW]     reference blocks (a, b), a <= b used: LinearElastic 3 x 3 row-major + pad (GW = 10),
                         //              Laplace [G00, G01 + G10, G02 + G20, G11, G12 + G21, G22] (GW = 6)
    int rw, cs, us, nbs, npos, acc_max;
};

constexpr int AFFINE_GW_LE = 10, AFFINE_GW_LAP = 6;

// doubles per staging buffer: the rows of a block behind up to 15 carried values (see write_out)
__host__ __device__ inline int affine_stage_doubles(int acc_max) { return (acc_max + 16 + 1) & ~1; }

__host__ __device__ inline size_t affine_lds_bytes(int op, int us, int acc_max, int rw) {
    const int gw = (op == FH_LAPLACE) ? AFFINE_GW_LAP : AFFINE_GW_LE;
    const int accp = affine_stage_doubles(acc_max);
    return sizeof(double) * ((size_t)64 * gw + (size_t)2 * us * gw + (size_t)2 * accp) + sizeof(int) * ((size_t)2 * rw + 4);
}

// 1 if the trilinear map of a hexahedron is affine to the relative tolerance `tol`: with the node signs of
// hexahedron.rs:49-58 the map is  c0 + c1 xi + c2 eta + c3 zeta + c12 xi eta + c23 eta zeta + c31 zeta xi + c123 xi eta zeta,
// c_* = 1/8 sum_a sign X_a; affine iff the four mixed coefficients vanish.  Compared against the shortest of c1, c2, c3.
__global__ void __launch_bounds__(256) k_classify_affine_hex8(const double* verts, const int* conn, long long E, double tol,
                                                              unsigned char* out, unsigned long long* count) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = e < E;
    bool ok = false;
    if (in_range) {
    const int sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    double c[7][3];
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int i = 0; i < 3; ++i) c[k][i] = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const double* X = verts + (size_t)conn[e * 8 + g] * 3;
        const int s[7] = {sx[g], sy[g], sz[g], sx[g] * sy[g], sy[g] * sz[g], sz[g] * sx[g], sx[g] * sy[g] * sz[g]};
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int i = 0; i < 3; ++i) c[k][i] += s[k] * X[i];
    }
    double lin_min = 1e300, dev = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) lin_min = fmin(lin_min, sqrt(c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2]));
#pragma unroll
    for (int k = 3; k < 7; ++k) dev = fmax(dev, sqrt(c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2]));
    ok = lin_min > 0.0 && dev <= tol * lin_min;  // NaN / inf coordinates compare false: general path
    out[e] = ok ? 1 : 0;
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}

__global__ void __launch_bounds__(256) k_bytes_differ(const unsigned char* x, const unsigned char* y, long long n, int* differ) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && x[i] != y[i]) *differ = 1;
}

// class of a node block: 1 if every adjacent element is affine (and the block fits the affine kernel's slot budget)
__global__ void __launch_bounds__(256) k_block_class(const GatherHdr* hdr, const unsigned* gt_elems, const unsigned char* elem_aff,
                                                     int nblk, int max_u, unsigned char* cls) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const GatherHdr h = hdr[b];
    unsigned char ok = (h.U <= max_u) ? 1 : 0;
    for (int k = 0; k < h.U && ok; ++k) ok = elem_aff[gt_elems[h.u_off + k]];
    cls[b] = ok;
}

template <int OP>
__global__ void __launch_bounds__(256, 3) k_gather_affine(const KArgs a, const AffineTables T) {
    constexpr int D = 3;
    constexpr bool LAP = (OP == FH_LAPLACE);
    constexpr int S = LAP ? 1 : 3;
    constexpr int GW = LAP ? AFFINE_GW_LAP : AFFINE_GW_LE;
    constexpr int NV = LAP ? 1 : 9;  // values per output block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    double* GH = lds;                        // 64 * GW
    double* JS = GH + 64 * GW;               // [2][us][GW]
    const int accp = affine_stage_doubles(T.acc_max);
    double* OUT = JS + 2 * T.us * GW;        // [2][accp]   (even offsets: 16-byte aligned)
    int* lds_i = reinterpret_cast<int*>(OUT + 2 * accp);  // [2][rw]
    const int tid = threadIdx.x;
    const int G = gridDim.x, npos = T.npos;
    const int p_begin = (int)((long long)blockIdx.x * npos / G), p_end = (int)((long long)(blockIdx.x + 1) * npos / G);
    if (p_begin >= p_end) return;
    for (int i = tid; i < 64 * GW; i += 256) GH[i] = T.ghat[i];
    for (int i = tid; i < 2 * accp; i += 256) OUT[i] = 0.0;

    // vertex role: lane (slot vs, node pair vq) holds nodes vq and vq + 4 of its slot -- 4 us <= 128 lanes, waves 0 and 1
    const int vs = tid >> 2, vq = tid & 3;
    const bool vwave = (tid & ~63) < 4 * T.us;  // wave-uniform
    struct Rec { int w, c0, c1; };
    auto load_rec = [&](int p, Rec& r) {  // branch-free, clamped (see k_gather_pipelined)
        p = min(p, npos - 1);
        r.w = T.rec[(size_t)p * T.rw + min(tid, T.rw - 1)];
        const int ci = min(vs * 8 + vq, T.cs - 5);
        r.c0 = T.conn[(size_t)p * T.cs + ci];
        r.c1 = T.conn[(size_t)p * T.cs + ci + 4];
    };
    auto load_lane = [&](int p) { return T.lanes[(size_t)min(p, npos - 1) * 256 + tid]; };
    double X0[D], X1[D];
    auto load_verts = [&](const Rec& r) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            X0[c] = a.verts[(size_t)r.c0 * D + c];
            X1[c] = a.verts[(size_t)r.c1 * D + c];
        }
    };
    auto rec_base = [&](int parity) { return lds_i + parity * T.rw; };
    // Jacobian record of every slot of position p from the vertices in registers.  J = X Ghat(0)^T with the centre
    // gradients sign / 8 (hexahedron.rs:63-83 at xi = 0): sums over the slot's eight nodes -- two in the lane, the rest by
    // quad permutes.  LinearElastic: sqrt(|det J|) J^-1 = sign(det J) rsqrt(|det J|) adj(J) (nine doubles);
    // Laplace: M = |det J| J^-1 J^-T (six doubles).
    auto slot_records = [&](int p, int parity) {
        if (!vwave) return;
        const double sxi = (vq == 1 || vq == 2) ? 1.0 : -1.0, seta = (vq >= 2) ? 1.0 : -1.0;
        double J[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double s = X0[i] + X1[i], dz = X1[i] - X0[i];
            const double ps = dpp_quad<0xB1>(s), pdz = dpp_quad<0xB1>(dz);      // lane ^ 1: the xi neighbour
            const double sum1 = s + ps, dxi1 = sxi * (s - ps), dz1 = dz + pdz;
            const double psum1 = dpp_quad<0x4E>(sum1), pdxi1 = dpp_quad<0x4E>(dxi1), pdz1 = dpp_quad<0x4E>(dz1);  // lane ^ 2: eta
            J[i][0] = 0.125 * (dxi1 + pdxi1);
            J[i][1] = 0.125 * (seta * (sum1 - psum1));
            J[i][2] = 0.125 * (dz1 + pdz1);
        }
        const double detJ = det_small<D>(J);
        double R[D][D];
        if (detJ == 0.0) {  // try_inverse fails only for det == 0 exactly (elliptic.rs:401-404); empty slots are degenerate too
            if (vq == 0 && vs < T.us) {
                const int e = T.elem[(size_t)p * T.us + vs];
                if (e >= 0) report_singular(a.status, (long long)e);
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) R[i][j] = 0.0;
        } else {
            adj_scaled(J, copysign(rsqrt_newton(fabs(detJ)), detJ), R);
        }
        if (vq == 0 && vs < T.us) {
            double* o = JS + ((size_t)parity * T.us + vs) * GW;
            if constexpr (LAP) {
                double M[6];
                int k = 0;
#pragma unroll
                for (int c = 0; c < D; ++c)
#pragma unroll
                    for (int d = c; d < D; ++d, ++k) M[k] = R[c][0] * R[d][0] + R[c][1] * R[d][1] + R[c][2] * R[d][2];
#pragma unroll
                for (int h = 0; h < 3; ++h) { f64x2 v; v.x = M[2 * h]; v.y = M[2 * h + 1]; *reinterpret_cast<f64x2*>(o + 2 * h) = v; }
            } else {
                const double r9[10] = {R[0][0], R[0][1], R[0][2], R[1][0], R[1][1], R[1][2], R[2][0], R[2][1], R[2][2], 0.0};
#pragma unroll
                for (int h = 0; h < 5; ++h) { f64x2 v; v.x = r9[2 * h]; v.y = r9[2 * h + 1]; *reinterpret_cast<f64x2*>(o + 2 * h) = v; }
            }
        }
    };
    // Rows of a finished block: LDS -> global memory.  The write path wants whole, aligned 128-byte lines (measured with
    // scripts/ubench_fill.hip: 16-byte stores that start a wave off a line boundary reach 4.3 TB/s instead of 6.2, and every
    // line written in two parts costs about ten full ones), but a block's rows start and end anywhere.  So the staging buffer
    // is laid out from the line boundary below the block's first value (`head` doubles in), only complete lines are stored,
    // and when the next position continues these rows (positions are in CSR order) the incomplete last line is carried
    // into the head of the other buffer instead of being written.  Lane t of step k stores the 16-byte piece t + 256 k of
    // the buffer: every wave writes 1 KiB from a line boundary.
    auto write_out = [&](double* line0, double* buf, double* other, int lo, int hi, bool carry_out) {
        const int L = carry_out ? (hi & ~15) : hi;   // stored now: [lo, L); carried: [L, hi)
        f64x2* buf2 = reinterpret_cast<f64x2*>(buf);
        f64x2* out2 = reinterpret_cast<f64x2*>(line0);
        const f64x2 zero2 = {0.0, 0.0};
        for (int q = tid; 2 * q < L; q += 256) {
            const int d0 = 2 * q;
            if (d0 >= lo && d0 + 2 <= L) {
                const f64x2 v = buf2[q];
                if (a.overwrite) out2[q] = v;
                else { const f64x2 o = out2[q]; f64x2 r; r.x = o.x + v.x; r.y = o.y + v.y; out2[q] = r; }
                buf2[q] = zero2;
            } else {  // the ends of a run of positions: single doubles
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (d0 + h >= lo && d0 + h < L) {
                        if (a.overwrite) line0[d0 + h] = buf[d0 + h]; else line0[d0 + h] += buf[d0 + h];
                        buf[d0 + h] = 0.0;
                    }
            }
        }
        if (carry_out && tid < hi - L) { other[tid] = buf[L + tid]; buf[L + tid] = 0.0; }
    };

    int p = p_begin;
    Rec nxt;
    uint2 lane_cur;
    {
        Rec cur;
        load_rec(p, cur);
        load_verts(cur);
        lane_cur = load_lane(p);
        load_rec(p + 1, nxt);
        if (tid < T.rw) rec_base(0)[tid] = cur.w;
        slot_records(p, 0);
        asm volatile("" : "+v"(nxt.w), "+v"(nxt.c0), "+v"(nxt.c1), "+v"(lane_cur.x), "+v"(lane_cur.y));
    }
    __syncthreads();

    const unsigned gh_base = (unsigned)(unsigned long long)GH;
    int parity = 0;
    bool carry_in = false;
    for (; p < p_end; ++p, parity ^= 1) {
        const bool have_next = (p + 1) < p_end;
        const int* rec = rec_base(parity);
        const GatherHdr hc = *reinterpret_cast<const GatherHdr*>(rec);
        const int* noff_l = rec + 8 + T.us / 4;
        load_verts(nxt);  // lands while this block is computed
        Rec nn;
        load_rec(p + 2, nn);
        uint2 lane_nxt = load_lane(p + 1);

        // terms: G = sum over the lane's terms of  R^T Ghat_(lo, hi) R  (transposed when a > b)
        const unsigned w0 = lane_cur.x, w1 = lane_cur.y;
        const int nterms = (int)((w0 >> 28) & 3u), grp = (int)(w0 >> 30);
        const unsigned js_base = (unsigned)(unsigned long long)(JS + (size_t)parity * T.us * GW);
        double Gt[2][NV];
        unsigned pj[2], pg[2];
        bool tsw[2], diag[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned term = (w0 >> (14 * t)) & 0x3fffu;
            const unsigned ta = (term >> 8) & 7u, tb = (term >> 11) & 7u;
            tsw[t] = ta > tb;
            diag[t] = ta == tb;
            pj[t] = js_base + (term & 255u) * (GW * 8);
            pg[t] = gh_base + (min(ta, tb) * 8u + max(ta, tb)) * (GW * 8);
        }
        if constexpr (LAP) {
            f64x2 m[2][3], g[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                m[t][0] = lds_read_f64x2<0>(pj[t]); m[t][1] = lds_read_f64x2<16>(pj[t]); m[t][2] = lds_read_f64x2<32>(pj[t]);
                g[t][0] = lds_read_f64x2<0>(pg[t]); g[t][1] = lds_read_f64x2<16>(pg[t]); g[t][2] = lds_read_f64x2<32>(pg[t]);
            }
            lds_wait<0>();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                asm volatile("" : "+v"(m[t][0]), "+v"(m[t][1]), "+v"(m[t][2]), "+v"(g[t][0]), "+v"(g[t][1]), "+v"(g[t][2]));
                double s = g[t][0].x * m[t][0].x;
                s = fma(g[t][0].y, m[t][0].y, s);
                s = fma(g[t][1].x, m[t][1].x, s);
                s = fma(g[t][1].y, m[t][1].y, s);
                s = fma(g[t][2].x, m[t][2].x, s);
                s = fma(g[t][2].y, m[t][2].y, s);
                Gt[t][0] = s;
            }
        } else {
            // five ds_read_b128 per operand; term 1's fetches are in flight while term 0 is multiplied (15 operations
            // outstanding at most: the lgkm counter tracks 15)
            f64x2 r[2][5], g[2][5];
            auto fetch_r = [&](auto tk) {
                constexpr int t = decltype(tk)::value;
                r[t][0] = lds_read_f64x2<0>(pj[t]); r[t][1] = lds_read_f64x2<16>(pj[t]); r[t][2] = lds_read_f64x2<32>(pj[t]);
                r[t][3] = lds_read_f64x2<48>(pj[t]); r[t][4] = lds_read_f64x2<64>(pj[t]);
            };
            auto fetch_g = [&](auto tk) {
                constexpr int t = decltype(tk)::value;
                g[t][0] = lds_read_f64x2<0>(pg[t]); g[t][1] = lds_read_f64x2<16>(pg[t]); g[t][2] = lds_read_f64x2<32>(pg[t]);
                g[t][3] = lds_read_f64x2<48>(pg[t]); g[t][4] = lds_read_f64x2<64>(pg[t]);
            };
            auto sandwich = [&](auto tk) {
                constexpr int t = decltype(tk)::value;
                asm volatile("" : "+v"(r[t][0]), "+v"(r[t][1]), "+v"(r[t][2]), "+v"(r[t][3]), "+v"(r[t][4]));
                asm volatile("" : "+v"(g[t][0]), "+v"(g[t][1]), "+v"(g[t][2]), "+v"(g[t][3]), "+v"(g[t][4]));
                const double Rm[3][3] = {{r[t][0].x, r[t][0].y, r[t][1].x}, {r[t][1].y, r[t][2].x, r[t][2].y}, {r[t][3].x, r[t][3].y, r[t][4].x}};
                const double Gm[3][3] = {{g[t][0].x, g[t][0].y, g[t][1].x}, {g[t][1].y, g[t][2].x, g[t][2].y}, {g[t][3].x, g[t][3].y, g[t][4].x}};
                double Tm[3][3], H[3][3];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int s = 0; s < 3; ++s) Tm[c][s] = fma(Gm[c][2], Rm[2][s], fma(Gm[c][1], Rm[1][s], Gm[c][0] * Rm[0][s]));
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int s = 0; s < 3; ++s) H[i][s] = fma(Rm[2][i], Tm[2][s], fma(Rm[1][i], Tm[1][s], Rm[0][i] * Tm[0][s]));
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int s = 0; s < 3; ++s) Gt[t][3 * i + s] = tsw[t] ? H[s][i] : H[i][s];
            };
            fetch_r(std::integral_constant<int, 0>{});
            fetch_g(std::integral_constant<int, 0>{});
            fetch_r(std::integral_constant<int, 1>{});
            lds_wait<5>();
            fetch_g(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            sandwich(std::integral_constant<int, 0>{});
            lds_wait<0>();
            sandwich(std::integral_constant<int, 1>{});
        }
        // unused terms read slot 0 / block (0, 0) (valid memory, arbitrary contents): discard by selection
        const bool two_out = (w1 >> 20) & 1u;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (nterms < 1) Gt[0][k] = 0.0;
            if (nterms < 2) Gt[1][k] = 0.0;
            if (!two_out) Gt[0][k] += Gt[1][k];
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double t1 = dpp_quad<0xB1>(Gt[0][k]);
            if (grp >= 1) Gt[0][k] += t1;
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double t2 = dpp_quad<0x4E>(Gt[0][k]);
            if (grp >= 2) Gt[0][k] += t2;
        }
        double* acc = OUT + (size_t)parity * accp;
        // the block's first value sits `head` doubles behind a 128-byte line boundary of the output
        const size_t g0 = (size_t)S * S * (size_t)hc.r0;
        const int head = (int)(((reinterpret_cast<size_t>(a.vals) >> 3) + g0) & 15);
        auto stage_block = [&](const double (&Gm)[NV], int il, int pos, bool dg) {
            const int rb = noff_l[il], cnt = noff_l[il + 1] - rb;
            double* base = acc + head + S * S * rb + S * pos;
            if constexpr (LAP) {
                base[0] = Gm[0];
            } else {
                const double tr = Gm[0] + Gm[4] + Gm[8];
                double v[3][3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        v[i][j] = (i == j) ? fma(a.mu, tr + Gm[4 * i], a.lambda * Gm[4 * i]) : fma(a.mu, Gm[3 * j + i], a.lambda * Gm[3 * i + j]);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) base[i * S * cnt + j] = (dg && i > j) ? v[j][i] : v[i][j];
            }
        };
        if ((w1 >> 21) & 1u) stage_block(Gt[0], (int)((w1 >> 7) & 7u), (int)(w1 & 127u), diag[0]);
        if ((w1 >> 22) & 1u) stage_block(Gt[1], (int)((w1 >> 17) & 7u), (int)((w1 >> 10) & 127u), diag[1]);

        // records of the next block (its vertices have landed by now)
        if (have_next) {
            if (tid < T.rw) rec_base(parity ^ 1)[tid] = nxt.w;
            slot_records(p + 1, parity ^ 1);
        }
        // everything prefetched lands here, before the barrier: a wait behind the write-out would also wait for its stores
        asm volatile("" : "+v"(nn.w), "+v"(nn.c0), "+v"(nn.c1), "+v"(lane_nxt.x), "+v"(lane_nxt.y));
        nxt = nn;
        lane_cur = lane_nxt;
        lds_barrier();
        // the parked header of the next position tells whether it continues these rows
        const bool carry_out = have_next && reinterpret_cast<const GatherHdr*>(rec_base(parity ^ 1))->r0 == hc.r0 + hc.nrow;
        write_out(a.vals + g0 - head, acc, OUT + (size_t)(parity ^ 1) * accp, carry_in ? 0 : head, head + S * S * hc.nrow, carry_out);
        carry_in = carry_out;
    }
}

}  // namespace fenris_hip
