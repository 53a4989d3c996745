// General (non-affine) Hex8 form of the row-owner stiffness kernel: Laplace / uniform LinearElastic, eight-point rule.
//
// What it replaces: k_gather_pipelined<HEX8, ., QC = 8, JT = 2> (assemble_kernels.hpp), whose lanes own a (node, element) entry and
// add their 3 x 3 blocks into row accumulators with ds_add_f64 -- 72 wavefront-level LDS atomics per position, a third of that
// kernel's time.  Here, like in k_affine_rows (affine_rows.hip), a lane owns an OUTPUT block (owned node I, column node J)
// and sums its terms in registers:
//     K_IJ = sum_{e contains I, J} sum_q  C( g_a(e,I),q , g_b(e,J),q ),     g_n,q = sqrt(w_q |det J_q|) J_q^-T ghat_n(xi_q)
//     (elliptic.rs:398-432: K_e += w |det J| C(grad phi_I, grad phi_J);  operators.rs:176-188 upper triangle, util.rs:38-51 mirror;
//      materials.rs:108-118:  mu ((a . b) I + b a^T) + lambda a b^T;   laplace.rs:60-68:  a . b)
// Differences from the affine form: the Jacobian varies over the points, so the operand of a term (element slot, local a, local b)
// is not one record R but the sixteen vectors g_a,q and g_b,q (24 doubles each, contiguous: 12 x ds_read_b128), computed by
//  * phase B: one lane per (new slot, point): J = X Ghat^T from the slot's vertices, R = sign(det) sqrt(w) rsqrt(|det|) adj(J),
//    g_n = R^T ghat_n for the eight nodes, stored [slot][node][q][c].  Slots persist along a sweep chain (the pipelined kernel's
//    position tables: consecutive positions of a chain share half of their elements), only the new slots are computed;
//  * phase C: the row lanes (lane records of affine_rows.hip, built by the same builder): two terms per lane, H = sum_q g_a g_b^T in
//    registers, groups of 2 / 4 lanes meet by DPP quad permutes, the material, three runs of three doubles into the staged rows.
// Roles per workgroup: four row waves (phases B and C), one loader wave (every global load: position records, vertex indices,
// vertices, lane tables; two positions ahead), one store wave (streams the staged rows of the previous position to global memory
// as whole 128-byte lines while phase B of the next one runs).  Two barriers per position; no LDS atomics, no accumulators to clear.
// 79 KB of LDS (54 KB of it the gradients of 32 slots): two workgroups per CU.
//
// Exact symmetry and run-to-run reproducibility exactly as in affine_rows.hip: both owners of a node pair evaluate the block of the
// pair's smaller node from the same operand values in the same order; the owner of the larger node stores the transpose.
#include <hip/hip_runtime.h>

#include <atomic>
#include <thread>

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>

#include "hex8_rows.hpp"
#include "small_ops.hpp"

namespace fenris_hip {

constexpr int HR_VS = 208;                 // bytes per (slot, node) vector: 24 doubles [q][c] + 16 (13 x 16 bytes: vectors of sixteen
                                           // consecutive (slot, node) pairs start in sixteen different 16-byte bank groups)
constexpr int HR_SS = 8 * HR_VS;           // bytes per slot: 104 x 16, i.e. 8 mod 16 pieces -- the bank group of vector (slot, node) is
                                           // (13 node + 8 (slot & 1)) mod 16: vectors of different local nodes never meet in a bank,
                                           // vectors of the same local node only when their slots have the same parity (hex8_rows_tune_lanes
                                           // arranges the lanes around that)
constexpr int HR_G_BYTES = HEX8_ROWS_US * HR_SS;
constexpr unsigned HR_ZERO_G = 64u;        // lane record: gidx 64 = absent term

static __host__ __device__ inline int hr_accp(int acc_max) { return (acc_max + 16 + 1) & ~1; }

size_t hex8_rows_lds_bytes(int acc_max) {
    return (size_t)HR_G_BYTES + HR_VS + sizeof(double) * ((size_t)hr_accp(acc_max) + HEX8_ROWS_US * 24 + 8 * 26 + 8) + 2 * 256 * sizeof(uint2) +
           4 * 16 * sizeof(int);
}

template <int OP, bool OVERWRITE, bool DBG>
__global__ void __launch_bounds__(HEX8_ROWS_THREADS, 4) k_hex8_rows(const KArgs a, const Hex8RowTables T, const int ablate_arg) {
    constexpr bool LAP = (OP == FH_LAPLACE);
    constexpr int S = LAP ? 1 : 3, SS = S * S;
    const int ablate = DBG ? ablate_arg : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* G = smem;                                            // [32 slots][8 nodes] vectors of 24 doubles
    double* ZV = reinterpret_cast<double*>(G + HR_G_BYTES);    // a vector of zeros: the operand of an absent term
    const int accp = hr_accp(T.acc_max);
    double* OUT = ZV + HR_VS / 8;                              // [accp] staged rows of one position, laid out from the line boundary below
    double* X = OUT + accp;                                    // [32][8][3] vertex coordinates per slot
    double* TAB = X + HEX8_ROWS_US * 24;                       // [8 points][26] reference gradients [node][c] of the geometry map
    double* SQW = TAB + 8 * 26;                                // [8] sqrt(w_q)
    uint2* LT = reinterpret_cast<uint2*>(SQW + 8);             // [2][256] two lane tables (along a sweep chain the slots of the retained and of the
                                                               // new elements change roles from one position to the next: two tables alternate)
    int* RING = reinterpret_cast<int*>(LT + 512);              // [4][16] position records, entry p & 3 (written two positions ahead)

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npos = T.npos, Gd = gridDim.x;
    const int p_begin = (int)((long long)blockIdx.x * npos / Gd), p_end = (int)((long long)(blockIdx.x + 1) * npos / Gd);
    if (p_begin >= p_end) return;
    for (int i = tid; i < HR_VS / 8; i += HEX8_ROWS_THREADS) ZV[i] = 0.0;
    for (int i = tid; i < accp; i += HEX8_ROWS_THREADS) OUT[i] = 0.0;
    for (int i = tid; i < 8 * 26; i += HEX8_ROWS_THREADS) TAB[i] = (i % 26 < 24) ? a.ggeom[(i / 26) * 24 + i % 26] : 0.0;
    if (tid < 8) SQW[tid] = sqrt(a.qw[tid]);
    const size_t vals_w = reinterpret_cast<size_t>(a.vals) >> 3;
    auto head_of = [&](int r0) { return (int)((vals_w + (size_t)SS * (size_t)r0) & 15); };
    auto rfl = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
    // FENRIS_HIP_TRACE (instrumented instantiation): cycles per role in the two halves of a position and at the two barriers
    unsigned long long tr[4] = {0, 0, 0, 0}, tr_t = 0;
    const bool tracing = DBG && a.trace != nullptr;
    auto tr_start = [&]() { if (tracing) tr_t = __builtin_readcyclecounter(); };
    auto tr_barrier = [&](int k) {   // work since the last stamp -> tr[k], the wait at the barrier -> tr[k + 1]
        if (tracing) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            lds_barrier();
            const unsigned long long t2 = __builtin_readcyclecounter();
            tr[k] += t1 - tr_t; tr[k + 1] += t2 - t1; tr_t = t2;
        } else lds_barrier();
    };
    auto tr_report = [&](int role) {
        if (tracing && (tid & 63) == 0) {
            for (int k = 0; k < 4; ++k) atomicAdd(a.trace + 7 * role + k, tr[k]);
            atomicAdd(a.trace + 7 * role + 6, 1ull);
            if (role == 0) a.trace[30] = 0x48455838ull;
        }
    };

    if (wave == 5) {
        // ------------------------------------------------------------------------------------------ store wave
        // Rows of a finished position: LDS -> global memory, whole aligned 128-byte lines only (see affine_rows.hip: the buffer is laid
        // out from the line boundary below the block's first value; when the next position of this workgroup continues these rows in
        // memory the incomplete last line is carried to the head of the buffer instead of being written).  ONE buffer: this wave reads it
        // between the end of a position's phase C and the end of the next position's phase B, the row waves write it during phase C.
        const int lane = tid - 320;
        // (round 5: wave priorities with the launch, FENRIS_HIP_HEX8_PRIO = store | loader << 2 | phase-B row waves << 4)
        { const int pr = (ablate_arg >> HEX8_ROWS_PRIO_SHIFT) & 3; if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); }
        auto put = [&](f64x2* dst, f64x2 val) {
            if (DBG && (ablate & 1)) return;
            if constexpr (OVERWRITE) *dst = val;
            else { const f64x2 o = *dst; f64x2 r; r.x = o.x + val.x; r.y = o.y + val.y; *dst = r; }
        };
        auto put1 = [&](double* dst, double val) {
            if (DBG && (ablate & 1)) return;
            if constexpr (OVERWRITE) *dst = val; else *dst += val;
        };
        auto stream_out = [&](const int4 hv, bool carry_in, bool carry_out) {
            const int r0 = rfl(hv.x), nrow = rfl(hv.y), head = (rfl(hv.w) >> 16) & 15;
            double* line0 = a.vals + (size_t)SS * (size_t)r0 - head;
            const int lo = carry_in ? 0 : head, hi = head + SS * nrow;
            const int L = carry_out ? (hi & ~15) : hi;          // stored now: [lo, L); carried: [L, hi)
            const int k0 = (lo + 1) >> 1, k1 = L >> 1;           // whole 16-byte pieces [k0, k1)
            const int np = max(k1 - k0, 0);
            const int nfull = np / 64, rem = np - nfull * 64;
            f64x2* b2 = reinterpret_cast<f64x2*>(OUT) + k0 + lane;
            f64x2* gout = reinterpret_cast<f64x2*>(line0) + k0 + lane;
            int i = 0;
            for (; i + 4 <= nfull; i += 4) {
                const f64x2 v0 = b2[64 * i], v1 = b2[64 * (i + 1)], v2 = b2[64 * (i + 2)], v3 = b2[64 * (i + 3)];
                put(gout + 64 * i, v0); put(gout + 64 * (i + 1), v1); put(gout + 64 * (i + 2), v2); put(gout + 64 * (i + 3), v3);
            }
            for (; i < nfull; ++i) put(gout + 64 * i, b2[64 * i]);
            if (lane < rem) put(gout + 64 * nfull, b2[64 * nfull]);
            const int e_lo = ((lo & 1) && lo < L) ? lo : -1;     // the ends of a run of positions: single doubles
            const int e_hi = ((L & 1) && L - 1 >= lo) ? L - 1 : -1;
            if (e_lo >= 0 && lane == 0) put1(line0 + e_lo, OUT[e_lo]);
            if (e_hi >= 0 && lane == 0) put1(line0 + e_hi, OUT[e_hi]);
            if (carry_out && lane < hi - L) { const double c = OUT[L + lane]; OUT[lane] = c; }   // (L is a multiple of 16 > lane, or 0: in place)
        };
        lds_barrier();  // B0
        tr_start();
        bool carry_in = false;
        for (int p = p_begin; p < p_end; ++p) {
            if (p > p_begin) {
                const int4 h_prev = *reinterpret_cast<const int4*>(RING + 16 * ((p - 1) & 3));
                const int r0_cur = rfl(RING[16 * (p & 3)]);
                bool carry_out = r0_cur == rfl(h_prev.x) + rfl(h_prev.y);
                if (carry_out && (rfl(h_prev.z) & 8)) {
                    // a position whose rows end before the first line boundary behind their start has nothing to store now, and what it
                    // would hand on starts at the line's beginning, below its own first value: it stores its own piece itself
                    const int head_p = (rfl(h_prev.w) >> 16) & 15, lo_p = carry_in ? 0 : head_p, hi_p = head_p + SS * rfl(h_prev.y);
                    if ((hi_p & ~15) < lo_p) carry_out = false;
                }
                stream_out(h_prev, carry_in, carry_out);
                carry_in = carry_out;
            }
            tr_barrier(0);  // B1(p): phase B done, the staged rows of p - 1 read
            tr_barrier(2);  // B2(p): phase C done, the rows of p staged
        }
        stream_out(*reinterpret_cast<const int4*>(RING + 16 * ((p_end - 1) & 3)), carry_in, false);
        tr_report(3);
        return;
    }

    if (wave == 4) {
        { const int pr = (ablate_arg >> (HEX8_ROWS_PRIO_SHIFT + 2)) & 3; if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); }
        // ------------------------------------------------------------------------------------------ loader wave
        // Every global load of the kernel.  Position p is current between the barriers B2(p - 1) and B2(p); the loader works in the
        // second half of that interval (phase C: nobody reads X, the lane table or the ring entry it writes):
        //   parks   the vertices of p + 1's new slots (requested a position ago), the lane table of p + 1 if it differs,
        //   writes  the ring entry of p + 2 (its record was requested a position ago),
        //   requests the vertices of p + 2 (through the indices requested a position ago), the lane table of p + 2 if it differs,
        //           the record and the vertex indices of p + 3.
        // Lane l holds the four vertices (slot l / 2, local nodes 4 (l % 2) ...); slots that stay staged fetch vertex 0 (one line).
        const int lane = tid - 256;
        const int slot_l = lane >> 1;
        const int nint4 = T.cs >> 2;
        auto load_pos = [&](int p) { return T.pos[(size_t)(unsigned)min(p, npos - 1) * 4u + (unsigned)(lane & 3)]; };
        auto load_conn = [&](int p) { return reinterpret_cast<const int4*>(T.conn + (size_t)(unsigned)min(p, npos - 1) * (unsigned)T.cs)[min(lane, nint4 - 1)]; };
        auto load_tab = [&](int id, int half) { return reinterpret_cast<const uint4*>(T.lanes)[(size_t)(unsigned)id * 128u + 64u * half + lane]; };
        auto park_tab = [&](int buf, int half, uint4 v) { reinterpret_cast<uint4*>(LT + 256 * buf)[64 * half + lane] = v; };
        struct Verts { double v[4][3]; };
        auto load_verts = [&](const int4 cn, unsigned mask) {
            Verts o;
            const bool fresh = ((mask >> slot_l) & 1u) != 0u;
            const int idx[4] = {fresh ? cn.x : 0, fresh ? cn.y : 0, fresh ? cn.z : 0, fresh ? cn.w : 0};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) o.v[k][c] = a.verts[(size_t)(unsigned)idx[k] * 3u + c];
            return o;
        };
        auto park_verts = [&](const Verts& o, unsigned mask) {
            if ((mask >> slot_l) & 1u) {
                double* dst = X + slot_l * 24 + (lane & 1) * 12;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int c = 0; c < 3; ++c) dst[3 * k + c] = o.v[k][c];
            }
        };
        // record words 0..3 live in the lanes with lane % 4 == 0, 12..15 in those with lane % 4 == 3
        auto ring_put = [&](int p, const int4 pr, int buf, bool changed, bool first) {
            if (lane < 4) {
                int4 o = pr;
                if (lane == 0) {
                    o.z = (pr.z & ~6) | (buf << 1) | (changed ? 4 : 0);
                    o.w = (first ? ((pr.w >> 8) & 0xff) : (pr.w & 0xff)) | (head_of(pr.x) << 16);
                }
                reinterpret_cast<int4*>(RING + 16 * (p & 3))[lane] = o;
            }
        };
        auto rl = [](int v, int l) { return __builtin_amdgcn_readlane(v, l); };
        // prologue: positions p_begin (every occupied slot is new to this workgroup) and p_begin + 1
        int4 posr = load_pos(p_begin);
        int4 pos1 = load_pos(p_begin + 1);
        int4 cn = load_conn(p_begin);
        const int4 cn1 = load_conn(p_begin + 1);
        int id_prev = rl(posr.z, 0) >> 8;
        int idbuf0 = id_prev, idbuf1 = -1;     // tables resident in the two buffers
        {
            const uint4 t0 = load_tab(id_prev, 0), t1 = load_tab(id_prev, 1);
            park_tab(0, 0, t0); park_tab(0, 1, t1);
        }
        ring_put(p_begin, posr, 0, true, true);
        {
            const unsigned occ0 = (unsigned)rl(posr.y, 3);
            const Verts v0 = load_verts(cn, occ0);
            park_verts(v0, occ0);
        }
        const int id1 = rl(pos1.z, 0) >> 8;
        bool tab_pending = id1 != id_prev;
        int buf_prev = tab_pending ? 1 : 0, buf_pending = 1;    // buffer of position p + 1 while p is current; where the pending table goes
        ring_put(p_begin + 1, pos1, buf_prev, tab_pending, false);
        uint4 tab0 = {0, 0, 0, 0}, tab1 = {0, 0, 0, 0};
        if (tab_pending) { tab0 = load_tab(id1, 0); tab1 = load_tab(id1, 1); idbuf1 = id1; }
        id_prev = id1;
        unsigned mask_nxt = (unsigned)rl(pos1.x, 3);          // new slots of position p + 1 while p is current
        Verts vc = load_verts(cn1, mask_nxt);
        cn = load_conn(p_begin + 2);
        posr = load_pos(p_begin + 2);
        lds_barrier();  // B0
        tr_start();
        for (int p = p_begin; p < p_end; ++p) {
            tr_barrier(0);  // B1(p)
            park_verts(vc, mask_nxt);                         // X of p + 1
            if (tab_pending) { park_tab(buf_pending, 0, tab0); park_tab(buf_pending, 1, tab1); }
            const int id2 = rl(posr.z, 0) >> 8;
            const bool ch2 = id2 != id_prev;
            // the table of p + 2: resident in one of the two buffers, or fetched into the one that p + 1 does not use
            const bool fetch2 = id2 != idbuf0 && id2 != idbuf1;
            const int buf2 = fetch2 ? (buf_prev ^ 1) : (id2 == idbuf0 ? 0 : 1);
            ring_put(p + 2, posr, buf2, ch2, false);
            mask_nxt = (unsigned)rl(posr.x, 3);
            if (!(DBG && (ablate & 16))) {                    // (profiling: no global loads in the sweep)
                vc = load_verts(cn, mask_nxt);                // vertices of p + 2
                tab_pending = fetch2;
                if (fetch2) {
                    tab0 = load_tab(id2, 0); tab1 = load_tab(id2, 1);
                    buf_pending = buf2;
                    if (buf2 == 0) idbuf0 = id2; else idbuf1 = id2;
                }
                buf_prev = buf2;
                id_prev = id2;
                cn = load_conn(p + 3);
                posr = load_pos(p + 3);
            }
            tr_barrier(2);  // B2(p)
        }
        tr_report(2);
        return;
    }

    // ---------------------------------------------------------------------------------------------- row waves
    if (wave >= 2) { const int pr = (ablate_arg >> (HEX8_ROWS_PRIO_SHIFT + 4)) & 3; if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); }
    lds_barrier();  // B0
    tr_start();
    const unsigned g_addr = (unsigned)(unsigned long long)G, zv_addr = (unsigned)(unsigned long long)ZV;
    const unsigned idle_x = (HR_ZERO_G << 5) | (HR_ZERO_G << 17);
    uint2 lane_cur = {idle_x, 0u};
    bool wave_works = true;
    // phase B on the LAST row lanes: the first two row waves share their SIMDs with the loader and the store wave
    const int bt = 255 - tid;
    for (int p = p_begin; p < p_end; ++p) {
        const int* ring = RING + 16 * (p & 3);
        const int zs = rfl(ring[2]), ws = rfl(ring[3]);
        const int nnew = ws & 0xff, head = (ws >> 16) & 15;
        if (zs & 4) {   // the lane table changed with this position
            lane_cur = LT[256 * ((zs >> 1) & 1) + tid];
            wave_works = __builtin_amdgcn_ballot_w64(lane_cur.x != idle_x) != 0ull;   // (a wavefront without a lane skips phase C)
        }
        // ------------------------------------------------------------------ phase B: gradients of the new slots
        // One lane per (new slot, point).  (Measured and dropped: two lanes per item when at most sixteen slots are new -- both evaluate
        // the Jacobian, each writes four of the eight nodes, all four row waves share the phase: 9.6 -> 10.9 ms.  The CU is bound by the
        // total LDS and fp64 work of its two workgroups, not by the length of this phase.)
        const bool split = false;
        const int it = bt;
        if (it < nnew * 8 && !(DBG && (ablate & 4))) {
            const int q = it & 7;
            const int slot = (int)reinterpret_cast<const unsigned char*>(ring + 4)[it >> 3];
            const f64x2* xs = reinterpret_cast<const f64x2*>(X + slot * 24);
            const f64x2* tb = reinterpret_cast<const f64x2*>(TAB + q * 26);
            double xd[24], gd[24];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const f64x2 xv = xs[k], gv = tb[k];
                xd[2 * k] = xv.x; xd[2 * k + 1] = xv.y; gd[2 * k] = gv.x; gd[2 * k + 1] = gv.y;
            }
            double J[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) J[i][j] = xd[i] * gd[j];
#pragma unroll
            for (int n = 1; n < 8; ++n)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) J[i][j] = fma(xd[3 * n + i], gd[3 * n + j], J[i][j]);
            const double detJ = det_small<3>(J);
            double R[3][3];
            if (detJ == 0.0) {   // try_inverse fails only for det == 0 exactly (elliptic.rs:401-404)
                report_singular(a.status, (long long)T.elem[(size_t)(unsigned)p * (unsigned)T.us + (unsigned)slot]);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) R[i][j] = 0.0;
            } else {
                // sqrt(w |det J|) J^-1 = sign(det J) sqrt(w) rsqrt(|det J|) adj(J)
                adj_scaled(J, copysign(SQW[q], detJ) * rsqrt_newton(fabs(detJ)), R);
            }
            double* gs = reinterpret_cast<double*>(G + slot * HR_SS) + 3 * q;
            for (int pass = 0; pass < (split ? 1 : 2); ++pass) {   // (scalar trip count)
                const bool upper = split ? ((bt & 1) != 0) : (pass != 0);   // nodes 4 .. 7
                double* o = gs + (upper ? 4 : 0) * (HR_VS / 8);
#pragma unroll
                for (int n = 0; n < 4; ++n, o += HR_VS / 8) {
                    const double g0 = upper ? gd[12 + 3 * n] : gd[3 * n], g1 = upper ? gd[13 + 3 * n] : gd[3 * n + 1], g2 = upper ? gd[14 + 3 * n] : gd[3 * n + 2];
#pragma unroll
                    for (int i = 0; i < 3; ++i) o[i] = fma(R[2][i], g2, fma(R[1][i], g1, R[0][i] * g0));
                }
            }
        }
        tr_barrier(0);  // B1(p)

        // ------------------------------------------------------------------ phase C: the lane's block
        const unsigned x = lane_cur.x, y = lane_cur.y;
        const unsigned gi0 = (x >> 5) & 127u, gi1 = (x >> 17) & 127u;
        const unsigned b0 = g_addr + (x & 31u) * HR_SS, b1 = g_addr + ((x >> 12) & 31u) * HR_SS;
        const bool z0 = gi0 == HR_ZERO_G, z1 = gi1 == HR_ZERO_G;
        unsigned pa0 = z0 ? zv_addr : b0 + (gi0 >> 3) * HR_VS, pb0 = z0 ? zv_addr : b0 + (gi0 & 7u) * HR_VS;
        unsigned pa1 = z1 ? zv_addr : b1 + (gi1 >> 3) * HR_VS, pb1 = z1 ? zv_addr : b1 + (gi1 & 7u) * HR_VS;
        if (DBG && (ablate & 8)) {   // (profiling: sixteen consecutive vectors per sixteen lanes -- no bank conflicts, wrong sums)
            pa0 = g_addr + (unsigned)(tid & 15) * HR_VS; pb0 = g_addr + (unsigned)((tid + 5) & 15) * HR_VS;
            pa1 = g_addr + HR_SS + (unsigned)((tid + 3) & 15) * HR_VS; pb1 = g_addr + HR_SS + (unsigned)((tid + 11) & 15) * HR_VS;
        }
        const int grp = (int)((x >> 24) & 3u);
        // one accumulator per term: H = H0 + H1 does not depend on which of the lane's two terms sits in which half of the record (the
        // lane tuner swaps them), and the owners of (I, J) and (J, I) hold the same pairs of terms
        double H[3][3], H1[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { H[i][j] = 0.0; H1[i][j] = 0.0; }
        double sl = 0.0, sl1 = 0.0;
        if (wave_works && !(DBG && (ablate & 2)) && !(DBG && (ablate & 32) && wave == 3)) {   // (32, profiling: three row waves)
            // eight groups of two points: three 16-byte pieces of each operand vector; two groups in flight (12 of the 15 LDS
            // operations the counter tracks)
            f64x2 A[2][3], B[2][3];
            auto fetch = [&](auto gk) {
                constexpr int g = decltype(gk)::value, sb = g & 1, kk = g & 3;
                const unsigned pa = (g < 4) ? pa0 : pa1, pb = (g < 4) ? pb0 : pb1;
                A[sb][0] = lds_read_f64x2<(3 * kk) * 16>(pa);
                A[sb][1] = lds_read_f64x2<(3 * kk + 1) * 16>(pa);
                A[sb][2] = lds_read_f64x2<(3 * kk + 2) * 16>(pa);
                B[sb][0] = lds_read_f64x2<(3 * kk) * 16>(pb);
                B[sb][1] = lds_read_f64x2<(3 * kk + 1) * 16>(pb);
                B[sb][2] = lds_read_f64x2<(3 * kk + 2) * 16>(pb);
            };
            auto consume = [&](auto gk) {
                constexpr int g = decltype(gk)::value, sb = g & 1;
                const double ga[2][3] = {{A[sb][0].x, A[sb][0].y, A[sb][1].x}, {A[sb][1].y, A[sb][2].x, A[sb][2].y}};
                const double gb[2][3] = {{B[sb][0].x, B[sb][0].y, B[sb][1].x}, {B[sb][1].y, B[sb][2].x, B[sb][2].y}};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if constexpr (LAP) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) { if constexpr (g < 4) sl = fma(ga[t][i], gb[t][i], sl); else sl1 = fma(ga[t][i], gb[t][i], sl1); }
                    } else {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                if constexpr (g < 4) H[i][j] = fma(ga[t][i], gb[t][j], H[i][j]);
                                else H1[i][j] = fma(ga[t][i], gb[t][j], H1[i][j]);
                            }
                    }
                }
            };
            fetch(std::integral_constant<int, 0>{});
            fetch(std::integral_constant<int, 1>{});
            lds_wait<6>(); consume(std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0);
            fetch(std::integral_constant<int, 2>{});
            lds_wait<6>(); consume(std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0);
            fetch(std::integral_constant<int, 3>{});
            lds_wait<6>(); consume(std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
            fetch(std::integral_constant<int, 4>{});
            lds_wait<6>(); consume(std::integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0);
            fetch(std::integral_constant<int, 5>{});
            lds_wait<6>(); consume(std::integral_constant<int, 4>{}); __builtin_amdgcn_sched_barrier(0);
            fetch(std::integral_constant<int, 6>{});
            lds_wait<6>(); consume(std::integral_constant<int, 5>{}); __builtin_amdgcn_sched_barrier(0);
            fetch(std::integral_constant<int, 7>{});
            lds_wait<6>(); consume(std::integral_constant<int, 6>{}); __builtin_amdgcn_sched_barrier(0);
            lds_wait<0>(); consume(std::integral_constant<int, 7>{});
        }
        char* out_b = reinterpret_cast<char*>(OUT);
        // the staged rows of the previous position must have left the buffer (the store wave is normally done long before)
        sl += sl1;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) H[i][j] += H1[i][j];
        if constexpr (LAP) {
            if (grp >= 1) sl += dpp_quad_full<0xB1>(sl);
            if (grp >= 2) sl += dpp_quad_full<0x4E>(sl);
            if ((x >> 28) & 1u) {
                double* o = reinterpret_cast<double*>(out_b) + head;
                o[y & 0x1fffu] = sl;
                if ((x >> 29) & 1u) o[(y >> 16) & 0x1fffu] = sl;     // the twin block (J, I): the same number
            }
        } else {
            if (grp >= 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) H[i][j] += dpp_quad_full<0xB1>(H[i][j]);
            }
            if (grp >= 2) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) H[i][j] += dpp_quad_full<0x4E>(H[i][j]);
            }
            if ((x >> 28) & 1u) {
                // K = mu (tr H I + H^T) + lambda H for the block of the pair's smaller node; the owner of the larger node stores the
                // transpose, diagonal blocks mirror their upper triangle (util.rs:38-51)
                const bool tr = (x >> 26) & 1u, dg = (x >> 27) & 1u;
                const double mu_tr = a.mu * (H[0][0] + H[1][1] + H[2][2]);
                const double mpl = a.mu + a.lambda;
                double v[3][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) v[i][i] = fma(mpl, H[i][i], mu_tr);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = i + 1; j < 3; ++j) {
                        const double up = fma(a.mu, H[j][i], a.lambda * H[i][j]);   // (i, j)
                        const double lw = fma(a.mu, H[i][j], a.lambda * H[j][i]);   // (j, i)
                        v[i][j] = tr ? lw : up;
                        v[j][i] = (tr || dg) ? up : lw;
                    }
                // row strides: S x the column blocks of the node's rows (position record, bytes 56 ..)
                const unsigned char* cnts = reinterpret_cast<const unsigned char*>(ring + 14);
                const unsigned rs = 3u * cnts[(y >> 13) & 7u];
                double* stage = reinterpret_cast<double*>(out_b) + head + (y & 0x1fffu);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double* row = stage + i * rs;
                    row[0] = v[i][0]; row[1] = v[i][1]; row[2] = v[i][2];
                }
                if ((x >> 29) & 1u) {   // the twin block (J, I) of a pair of nodes this position owns both: the transpose
                    const unsigned rs2 = 3u * cnts[y >> 29];
                    double* st2 = reinterpret_cast<double*>(out_b) + head + ((y >> 16) & 0x1fffu);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        double* row = st2 + i * rs2;
                        row[0] = v[0][i]; row[1] = v[1][i]; row[2] = v[2][i];
                    }
                }
            }
        }
        tr_barrier(2);  // B2(p)
    }
    if (wave == 0) tr_report(0);
    if (wave == 3) tr_report(1);
}


// ------------------------------------------------------------------------------------------------ lane tuner (host)
// Bank model (MI355X_MICROARCH.md, LDS table): a ds_read_b128 of a wavefront is served in four groups of sixteen lanes
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same + 32); a group takes one LDS cycle per distinct address that falls on the same
// 16-byte bank group; equal addresses are broadcast.  Operand vector (slot, node) starts in bank group (13 node + 8 (slot & 1)) mod 16
// and all twelve pieces of the vectors of a term are read with the same offsets, so the cost of a lane arrangement is, summed over
// the lane groups, the two halves of the record and the two operands, the largest number of distinct vectors in one bank group.
namespace {
struct TuneLane { unsigned x, y; };
inline int hw_group(int lane) {   // 0..15 over 256 lanes
    const int w = lane >> 6, l = lane & 31, hi = (lane >> 5) & 1;
    const bool first = (l < 4) || (l >= 12 && l < 16) || (l >= 20 && l < 28);
    return w * 4 + hi * 2 + (first ? 0 : 1);
}
// cost of one lane group: sum over (half of the record, operand) of the largest number of distinct vectors sharing a bank group.  Operand
// vector (slot, node) lies in bank group (13 node + 8 (slot & 1)) mod 16 -- a bank group fixes the local node and the parity of the slot, so
// the vectors in it differ by slot >> 1: a 16-bit set (bit 16: the zero vector, which sits in bank group 0).
inline int group_cost(const TuneLane* L, const int* lanes16) {
    int total = 0;
    for (int t = 0; t < 2; ++t) {
        unsigned ma[16] = {0}, mb[16] = {0};
        for (int k = 0; k < 16; ++k) {
            const unsigned x = L[lanes16[k]].x;
            const unsigned slot = (x >> (12 * t)) & 31u, g = (x >> (5 + 12 * t)) & 127u;
            if (g == 64u) { ma[0] |= 1u << 16; mb[0] |= 1u << 16; continue; }
            const unsigned par8 = 8u * (slot & 1u), bit = 1u << (slot >> 1);
            ma[(13u * (g >> 3) + par8) & 15u] |= bit;
            mb[(13u * (g & 7u) + par8) & 15u] |= bit;
        }
        int xa = 0, xb = 0;
        for (int i = 0; i < 16; ++i) { xa = std::max(xa, __builtin_popcount(ma[i])); xb = std::max(xb, __builtin_popcount(mb[i])); }
        total += xa + xb;
    }
    return total;
}
// cost of the staging writes of sixteen CONTIGUOUS lanes (ds_write_b64 is served in four groups of sixteen contiguous lanes, banks of
// (address / 4) mod 32, i.e. the offset in doubles mod 16): the largest number of distinct offsets of storing lanes that share a bank.  The
// nine stores of a block (three rows of three doubles) shift every lane by the same amount, so one pattern stands for all nine.
inline int write_cost(const TuneLane* L, int first) {
    int offs[16], n = 0;
    for (int k = 0; k < 16; ++k) {
        const TuneLane& q = L[first + k];
        if (!((q.x >> 28) & 1u)) continue;
        offs[n++] = (int)(q.y & 0x1fffu);
    }
    int cnt[16] = {0}, mx = 0;
    for (int i = 0; i < n; ++i) {
        bool dup = false;
        for (int j = 0; j < i; ++j) dup |= offs[j] == offs[i];
        if (!dup) mx = std::max(mx, ++cnt[offs[i] & 15]);
    }
    return mx;
}
}  // namespace

void hex8_rows_tune_lanes(uint2* tables, int ntab, unsigned seed, double* cycles_before, double* cycles_after, long long total_proposals) {
    int members[16][16], fill[16] = {0};
    for (int l = 0; l < 256; ++l) { const int g = hw_group(l); members[g][fill[g]++] = l; }
    const int budget = std::max(500, std::min(40000, (int)(std::max(100000ll, total_proposals) / std::max(ntab, 1))));
    // the tables are independent: every table has its own random stream (seeded by its index: the result does not depend on the number of
    // threads), host threads take tables from a shared counter (3 M proposals: 0.8 s on one core for the 93 tables of a 216^3 mesh)
    std::vector<double> before_t((size_t)std::max(ntab, 1), 0.0), after_t((size_t)std::max(ntab, 1), 0.0);
    auto tune_one = [&](int tb) {
        unsigned long long rng = (0x9E3779B97F4A7C15ull ^ seed) + 0xD1B54A32D192ED03ull * (unsigned long long)(tb + 1);
        auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 11); };
        std::vector<TuneLane> best(256);
        double before = 0.0, after = 0.0;
        TuneLane* L = reinterpret_cast<TuneLane*>(tables) + (size_t)tb * 256;
        // total = 4 x (reads: twelve 16-byte pieces per unit) + 3 x (writes: nine 8-byte stores per unit)
        int gcost[16], wcost[16], total = 0, reads = 0;
        for (int g = 0; g < 16; ++g) { gcost[g] = group_cost(L, members[g]); wcost[g] = write_cost(L, 16 * g); reads += gcost[g]; total += 4 * gcost[g] + 3 * wcost[g]; }
        before += reads;
        int best_total = total;
        std::copy(L, L + 256, best.begin());
        // the lanes in use fill whole wavefronts from the front (a wavefront without a lane skips phase C): moves stay below `limit`
        int used = 0;
        for (int l = 0; l < 256; ++l)
            if (L[l].x != ((64u << 5) | (64u << 17))) used = l + 1;
        const int limit = std::max(64, std::min(256, (used + 63) & ~63));
        // unit size at every lane: 4 (aligned quad, log2(group) = 2), 2 (aligned pair) or 1
        auto unit_at = [&](int lane) { const unsigned g = (L[lane].x >> 24) & 3u; return g >= 2 ? 4 : g == 1 ? 2 : 1; };
        // simulated annealing over (a) swaps of two aligned blocks of 4 / 2 / 1 lanes that consist of whole units, (b) swaps of the two
        // halves of one lane's record
        for (int it = 0; it < budget; ++it) {
            const double temp = 4.0 * (1.0 - (double)it / budget) + 0.05;
            const unsigned r = rnd();
            const int kind = (int)(r & 3u);
            int sz = 0, a = 0, b = 0;
            unsigned saved_x = 0;
            if (kind == 0) {
                a = (int)((r >> 2) & 255u) % limit;
                const unsigned x = L[a].x, lo = x & 0xfffu, hi = (x >> 12) & 0xfffu;
                if (lo == hi) continue;
                saved_x = x;
                L[a].x = (x & 0xff000000u) | (lo << 12) | hi;
            } else {
                sz = kind == 1 ? 4 : kind == 2 ? 2 : 1;
                a = (int)((((r >> 2) & 255u) % limit) / sz) * sz;
                b = (int)((((r >> 10) & 255u) % limit) / sz) * sz;
                if (a == b) continue;
                bool ok = true;
                for (int base : {a, b}) {
                    if (sz < 4 && unit_at(base & ~3) == 4) ok = false;        // the block lies inside a quad
                    if (sz == 1 && unit_at(base & ~1) == 2) ok = false;       // ... inside a pair
                    for (int k = 0; k < sz && ok;) {
                        const int u = unit_at(base + k);
                        if (u > sz - k || ((base + k) % u) != 0) ok = false;  // a unit that sticks out of the block
                        k += u;
                    }
                }
                if (!ok) continue;
                for (int k = 0; k < sz; ++k) std::swap(L[a + k], L[b + k]);
            }
            int touched[8], nt = 0;
            auto touch = [&](int lane) {
                const int g = hw_group(lane);
                for (int i = 0; i < nt; ++i) if (touched[i] == g) return;
                touched[nt++] = g;
            };
            if (kind == 0) touch(a);
            else for (int k = 0; k < sz; ++k) { touch(a + k); touch(b + k); }
            int d = 0, old[8], wt[8], nw = 0, oldw[8];
            auto touchw = [&](int lane) {
                const int g = lane >> 4;
                for (int i = 0; i < nw; ++i) if (wt[i] == g) return;
                wt[nw++] = g;
            };
            if (kind != 0) for (int k = 0; k < sz; ++k) { touchw(a + k); touchw(b + k); }
            for (int i = 0; i < nt; ++i) { old[i] = gcost[touched[i]]; const int c = group_cost(L, members[touched[i]]); d += 4 * (c - old[i]); gcost[touched[i]] = c; }
            for (int i = 0; i < nw; ++i) { oldw[i] = wcost[wt[i]]; const int c = write_cost(L, 16 * wt[i]); d += 3 * (c - oldw[i]); wcost[wt[i]] = c; }
            const bool accept = d <= 0 || (double)(rnd() & 0xffffu) / 65536.0 < std::exp(-(double)d / temp);
            if (!accept) {
                if (kind == 0) L[a].x = saved_x;
                else for (int k = 0; k < sz; ++k) std::swap(L[a + k], L[b + k]);
                for (int i = 0; i < nt; ++i) gcost[touched[i]] = old[i];
                for (int i = 0; i < nw; ++i) wcost[wt[i]] = oldw[i];
            } else {
                total += d;
                if (total < best_total) { best_total = total; std::copy(L, L + 256, best.begin()); }
            }
        }
        std::copy(best.begin(), best.end(), L);
        { int r2 = 0; for (int g = 0; g < 16; ++g) r2 += group_cost(L, members[g]); after += r2; }
        before_t[(size_t)tb] = before;
        after_t[(size_t)tb] = after;
    };
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (int tb = next.fetch_add(1); tb < ntab; tb = next.fetch_add(1)) tune_one(tb);
    };
    const int nthreads = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, ntab}));
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread& th : pool) th.join();
    double before = 0.0, after = 0.0;
    for (int tb = 0; tb < ntab; ++tb) { before += before_t[(size_t)tb]; after += after_t[(size_t)tb]; }
    if (cycles_before) *cycles_before = ntab ? before / ntab : 0.0;
    if (cycles_after) *cycles_after = ntab ? after / ntab : 0.0;
}

// ------------------------------------------------------------------------------------------------ position records
__global__ void __launch_bounds__(256) k_hex8_rows_positions(const int* p_rec, int rw, int us, int ms, const int4* hdr, int npos, int4* pos) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int* rec = p_rec + (size_t)p * rw;
    const GatherHdr h = *reinterpret_cast<const GatherHdr*>(rec);   // k0 = number of new slots (PipeTables::rec)
    const unsigned char* list = reinterpret_cast<const unsigned char*>(rec + 8);
    const int4 hd = hdr[p];
    int w[8];
    for (int k = 0; k < 8; ++k) w[k] = (k < us / 4) ? rec[8 + k] : 0;
    unsigned fresh = 0u, occ = 0u;
    for (int li = 0; li < h.U && li < us; ++li) {
        occ |= 1u << list[li];
        if (li < h.k0) fresh |= 1u << list[li];
    }
    pos[(size_t)p * 4 + 0] = make_int4(hd.x, hd.y, hd.z, (h.k0 & 0xff) | ((h.U & 0xff) << 8));
    pos[(size_t)p * 4 + 1] = make_int4(w[0], w[1], w[2], w[3]);
    pos[(size_t)p * 4 + 2] = make_int4(w[4], w[5], w[6], w[7]);
    // column blocks per row of every node of the position (the row stride of its staged rows is S x that many doubles)
    const int* noff = rec + 8 + us / 4 + ms + ms * 8 / 4;
    unsigned cw[2] = {0u, 0u};
    for (int il = 0; il < h.nb && il < 8; ++il) cw[il >> 2] |= (unsigned)((noff[il + 1] - noff[il]) & 0xff) << (8 * (il & 3));
    pos[(size_t)p * 4 + 3] = make_int4((int)fresh, (int)occ, (int)cw[0], (int)cw[1]);
}

hipError_t hex8_rows_positions(hipStream_t stream, const int* p_rec, int rw, int us, int ms, const int4* hdr, int npos, int4* pos) {
    if (npos <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_hex8_rows_positions, dim3((npos + 255) / 256), dim3(256), 0, stream, p_rec, rw, us, ms, hdr, npos, pos);
    return hipGetLastError();
}

hipError_t hex8_rows_launch(int op, int grid, size_t lds_bytes, hipStream_t stream, const KArgs& a, const Hex8RowTables& T, int ablate) {
    // (the instrumented instantiation exists for overwriting assemblies only: an accumulating one under FENRIS_HIP_TRACE / FENRIS_HIP_ABLATE runs
    // the production kernel -- it used to take the instrumented one and OVERWRITE the values)
    const bool ow = a.overwrite != 0, dbg = (ablate & 0x1ffff) != 0 && ow;
    if (!dbg) ablate &= ~0x1ffff;   // (the priority bits stay)
    void (*kern)(const KArgs, const Hex8RowTables, int);
    if (op == FH_LAPLACE) kern = dbg ? k_hex8_rows<FH_LAPLACE, true, true> : ow ? k_hex8_rows<FH_LAPLACE, true, false> : k_hex8_rows<FH_LAPLACE, false, false>;
    else kern = dbg ? k_hex8_rows<FH_LINEAR_ELASTIC, true, true> : ow ? k_hex8_rows<FH_LINEAR_ELASTIC, true, false> : k_hex8_rows<FH_LINEAR_ELASTIC, false, false>;
    if (lds_bytes > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(HEX8_ROWS_THREADS), lds_bytes, stream, a, T, ablate);
    return hipGetLastError();
}

}  // namespace fenris_hip
