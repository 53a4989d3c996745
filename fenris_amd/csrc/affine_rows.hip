// Affine-element form of the owner-computes stiffness kernel (Hex8; Laplace / uniform LinearElastic), second version.
//
// On an element whose trilinear map is affine (a parallelepiped) the Jacobian is constant, so the quadrature loop of
// elliptic.rs:398-432 collapses: with  Ghat_ab = sum_q w_q ghat_a(xi_q) ghat_b(xi_q)^T  (reference gradients only, built once
// per quadrature table on the host) the block of the node pair (a, b) of an element is
//     H_ab = R^T Ghat_ab R,  R = sqrt(|det J|) J^-1      (= |det J| J^-T Ghat_ab J^-1)
//     K_ab = mu (tr H_ab I + H_ab^T) + lambda H_ab       (materials.rs:108-118 summed over the points)
//     K_ab = tr H_ab = <Ghat_ab, R R^T>                   (laplace.rs:60-68)
// -- no per-point Jacobians, no physical gradients, no q-loop.  Which elements qualify is decided per element from the
// vertex coordinates (k_classify_affine_hex8 in affine_kernel.hpp); node blocks all of whose elements qualify run here, the
// others keep the general kernels.
//
// Work distribution (one workgroup = 4 row waves + 1 loader wave + 1 store wave, persistent over a contiguous range of
// positions; position = block of up to seven consecutive nodes):
//  * row lanes: a lane owns an output block (owned node I, column node J) and evaluates up to two of its terms
//    (element slot, local a, local b); blocks with more terms are split over 2 / 4 adjacent lanes whose partial sums meet by
//    DPP quad permutes (self block of a structured mesh: 8 terms, face 4, edge 2, corner 1: 36 lanes per node, 252 per 7-node
//    block).  The finished 3 x 3 block goes to a staging buffer in LDS laid out like the CSR rows.  No LDS atomics, and no
//    global memory traffic at all: everything the row waves read comes through LDS.
//  * the loader wave issues every global load: the element records R (or M) of the next position's slots -- written per
//    element by k_affine_records right before this kernel --, the lane table when it changes, the position headers.
//  * the store wave streams the staged rows of the PREVIOUS position to global memory as 16-byte stores, whole 128-byte
//    lines only (an incomplete last line is carried into the next position's buffer).  It never loads from global memory,
//    so it never waits for its stores to drain (loads and stores share vmcnt): the stores of several positions stay in
//    flight.
// One barrier per position; staging buffers, slot records and lane tables are double-buffered, the headers ride a ring.
//
// Exact symmetry (util.rs:38-51 mirrors the upper triangle): the owners of (I, J) and (J, I) both evaluate the block of
// the pair with the smaller global node first (gidx below), the terms of a block are ordered by element id, the split over
// lanes and the DPP tree depend only on the number of terms, and the owner of the larger node stores the transpose -- so
// both add bitwise identical numbers in the same order.  Diagonal blocks mirror their upper triangle.  Run-to-run
// reproducible for the same reason.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "affine_rows.hpp"
#include "small_ops.hpp"

namespace fenris_hip {

// lane record:  x = slot0 | gidx0 << 5 | slot1 << 12 | gidx1 << 17 | log2(group) << 24 | transpose << 26 | diagonal << 27 | store << 28
//               y = byte offset of the block's first value in the staged rows | byte stride between its rows << 16
// gidx = 8 a + b selects Ghat_ab; gidx 64 is a block of zeros (absent term).
constexpr unsigned AR_ZERO_G = 64u;

size_t affine_rows_lds_bytes(int op, int us, int acc_max) {
    const int gw = (op == FH_LAPLACE) ? AFFINE_ROWS_GW_LAP : AFFINE_ROWS_GW_LE;
    const size_t accp = (size_t)((acc_max + 16 + 1) & ~1);
    return sizeof(double) * ((size_t)65 * gw + (size_t)2 * us * gw + 2 * accp) + 4 * sizeof(int4) + 2 * 256 * sizeof(uint2) + 64;   // (+ 64: where the lanes without a block write)
}

template <int OP, bool OVERWRITE, bool DBG, int DEPTH, int NSTORE, bool MASKED>
__global__ void __launch_bounds__(320 + 64 * NSTORE, 5)
k_affine_rows(const KArgs a, const AffineRowTables T, const int ablate_arg) {
    constexpr int NT = 320 + 64 * NSTORE;
    constexpr bool LAP = (OP == FH_LAPLACE);
    constexpr int S = LAP ? 1 : 3, SS = S * S;
    constexpr int GW = LAP ? AFFINE_ROWS_GW_LAP : AFFINE_ROWS_GW_LE;
    const int ablate = DBG ? (ablate_arg & 0xffff) : 0;
    const bool nt_stores = (ablate_arg & AFFINE_ROWS_NT_STORES) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* GH = reinterpret_cast<double*>(smem);   // [65][GW]
    double* JS = GH + 65 * GW;                      // [2][us][GW]
    const int accp = (T.acc_max + 16 + 1) & ~1;
    double* OUT = JS + 2 * T.us * GW;               // [2][accp]
    int4* HDR = reinterpret_cast<int4*>(OUT + 2 * accp);  // [4] ring of position headers, entry q & 3 = {first value, rows,
                                                    // flags | table slot << 1 | table changed << 2 | table id << 8, head}:
                                                    // written by the loader wave two positions ahead, read by every wave
    uint2* LT = reinterpret_cast<uint2*>(HDR + 4);  // [2][256] lane tables: the one in use and the one that comes next
    double* DUMP = reinterpret_cast<double*>(LT + 2 * 256);   // [8] Laplace row loop: the lanes that own no block write here instead of branching around the store

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, npos = T.npos_all;
    const int p_begin = T.pos0 + (int)((long long)blockIdx.x * T.npos / G);
    const int p_end = T.pos0 + (int)((long long)(blockIdx.x + 1) * T.npos / G);
    if (p_begin >= p_end) return;
    for (int i = tid; i < 65 * GW; i += NT) GH[i] = (i < 64 * GW) ? T.ghat[i] : 0.0;
    for (int i = tid; i < 2 * accp; i += NT) OUT[i] = 0.0;
    const size_t vals_w = reinterpret_cast<size_t>(a.vals) >> 3;
    auto head_of = [&](int r0) { return (int)((vals_w + (size_t)SS * (size_t)r0) & 15); };
    // FENRIS_HIP_TRACE (instrumented instantiation): cycles per role between barriers / at the barriers, summed over workgroups
    unsigned long long tr_work = 0, tr_bar = 0, tr_t0 = 0, tr_seg = 0;
    auto tr_start = [&]() { if (DBG && a.trace) tr_t0 = __builtin_readcyclecounter(); };
    auto tr_barrier = [&]() {
        if (DBG && a.trace) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            lds_barrier();
            const unsigned long long t2 = __builtin_readcyclecounter();
            tr_work += t1 - tr_t0; tr_bar += t2 - t1; tr_t0 = t2;
        } else lds_barrier();
    };
    auto tr_report = [&](int role) {
        if (DBG && a.trace && (tid & 63) == 0) {
            atomicAdd(a.trace + 7 * role + 0, tr_work); atomicAdd(a.trace + 7 * role + 1, tr_seg); atomicAdd(a.trace + 7 * role + 2, tr_bar); atomicAdd(a.trace + 7 * role + 6, 1ull);
        }
    };

    if (wave >= 5) {
        // ------------------------------------------------------------------------------------------ store wave(s)
        // NSTORE wavefronts share the work as one unit of SL = 64 NSTORE lanes: a trip moves SL consecutive 16-byte pieces.
        constexpr int SL = 64 * NSTORE;
        const int lane = tid - 320;
        // the store wave's own instruction stream is the critical path of a position (78 % of it busy): its instructions go first on its SIMD
        // (round 5; the level comes with the launch -- FENRIS_HIP_AFFINE_PRIO = store | loader << 2, default 3 | 2 << 2 -- one scalar branch per role and launch)
        { const int pr = (ablate_arg >> AFFINE_ROWS_PRIO_SHIFT) & 3; if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); }
        // Rows of a finished position: LDS -> global memory.  The write path wants whole, aligned 128-byte lines (16-byte
        // stores that start a wave off a line boundary reach 4.3 TB/s instead of 6.2, and a line written in two parts costs
        // about ten full ones: scripts/ubench_fill.hip), but a block's rows start and end anywhere.  So the staging buffer is
        // laid out from the line boundary below the block's first value (`head` doubles in), only complete lines are stored,
        // and when the next position continues these rows (positions are in CSR order) the incomplete last line is carried
        // into the head of the other buffer instead of being written.
        // The wave's own instruction stream is on the critical path from barrier to barrier, so everything about a position
        // is kept in scalar registers and the trips of the unrolled loops are skipped by scalar branches.
        auto rfl = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
        auto put = [&](f64x2* dst, f64x2 val) {
            if (DBG && (ablate & 1)) return;
            // non-temporal stores (Laplace only -- measured; the switch is gone): the rows are written once and never read by this
            // kernel.  3 % on Laplace; on elasticity equal within the run-to-run spread.
            if constexpr (OVERWRITE) { if (nt_stores) __builtin_nontemporal_store(val, dst); else *dst = val; }
            else { const f64x2 o = *dst; f64x2 r; r.x = o.x + val.x; r.y = o.y + val.y; *dst = r; }
        };
        auto put1 = [&](double* dst, double val) {
            if (DBG && (ablate & 1)) return;
            if constexpr (OVERWRITE) { if (nt_stores) __builtin_nontemporal_store(val, dst); else *dst = val; } else *dst += val;
        };
        auto stream_out = [&](const int4 hv, double* buf, double* other, bool carry_in, bool carry_out) {
            const int r0 = rfl(hv.x), nrow = rfl(hv.y), flags = rfl(hv.z), head = rfl(hv.w) & 15;
            double* line0 = a.vals + (size_t)SS * (size_t)r0 - head;
            const int lo = carry_in ? 0 : head, hi = head + SS * nrow;
            const int L = carry_out ? (hi & ~15) : hi;          // stored now: [lo, L); carried: [L, hi)
            (void)flags;
            const int k0 = (lo + 1) >> 1, k1 = L >> 1;           // whole 16-byte pieces [k0, k1)
            const int np = max(k1 - k0, 0);
            const int nfull = np / SL, rem = np - nfull * SL;    // trips of SL pieces, pieces of the last trip
            f64x2* b2 = reinterpret_cast<f64x2*>(buf) + k0 + lane;
            f64x2* gout = reinterpret_cast<f64x2*>(line0) + k0 + lane;
            int i = 0;
            for (; i + 4 <= nfull; i += 4) {
                const f64x2 v0 = b2[SL * i], v1 = b2[SL * (i + 1)], v2 = b2[SL * (i + 2)], v3 = b2[SL * (i + 3)];
                put(gout + SL * i, v0); put(gout + SL * (i + 1), v1); put(gout + SL * (i + 2), v2); put(gout + SL * (i + 3), v3);
            }
            for (; i < nfull; ++i) put(gout + SL * i, b2[SL * i]);
            if (lane < rem) put(gout + SL * nfull, b2[SL * nfull]);
            const int e_lo = ((lo & 1) && lo < L) ? lo : -1;     // the ends of a run of positions: single doubles
            const int e_hi = ((L & 1) && L - 1 >= lo) ? L - 1 : -1;
            if (e_lo >= 0 && lane == 0) put1(line0 + e_lo, buf[e_lo]);
            if (e_hi >= 0 && lane == 0) put1(line0 + e_hi, buf[e_hi]);
            if (carry_out && lane < hi - L) other[lane] = buf[L + lane];
            if constexpr (!MASKED) {
                // (Never taken: without a mask every block has an owner lane, bit 0 of the flags is always set.  The branch is what
                // remains of the clearing that used to live here; taking it OUT makes the unmasked sweep 3 - 5 % slower -- the
                // instruction stream of this wave is the critical path and the compiler lays it out differently -- measured with the
                // two builds side by side on one box, scripts/bin/ab_lib.sh: C2 0.291 / 0.307 ms, headline 4.70 / 4.87 ms.)
                if (!(flags & 1)) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const f64x2 z2 = {0.0, 0.0};
                    for (int t = 0; t < nfull; ++t) b2[SL * t] = z2;
                    if (lane < rem) b2[SL * nfull] = z2;
                    if (lane == 0 && e_lo >= 0) buf[e_lo] = 0.0;
                    if (lane == 0 && e_hi >= 0) buf[e_hi] = 0.0;
                    if (lane < hi - L) buf[L + lane] = 0.0;
                }
            }
        };
        if constexpr (LAP && !DBG && NSTORE == 1) {
            // Laplace / mass: a position's rows are at most a few hundred doubles, and what a position costs is the CU's ONE scalar unit
            // (profiles/r06_c2_lds_bound.txt: a scalar instruction costs ten vector ones here; the loop above runs ~110 of them per position).  The same
            // stream with everything that is uniform kept in VECTOR registers -- no readfirstlane, the ranges as per-lane predicates -- and the two
            // buffers addressed by one xor-toggled offset: ~25 scalar instructions per position.  Non-temporal stores always (see put).
            const int small_trips = (accp / 2 + 63) / 64;
            // (what a header holds is uniform, and the compiler knows: it would move every line below to the scalar unit -- `vec` hides that from it)
            auto vec = [](int4 h) { asm volatile("" : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w)); return h; };
            auto stream_small = [&](const int4 hv, unsigned buf_off, unsigned other_off, int cin, int cout) {
                const int head = hv.w & 15, hi = head + hv.y, lo = cin ? 0 : head;
                const int L = cout ? (hi & ~15) : hi;                  // stored now: [lo, L); carried: [L, hi)
                const int k0 = (lo + 1) >> 1, k1 = L >> 1;              // whole 16-byte pieces [k0, k1)
                double* line0 = a.vals + ((long long)hv.x - (long long)head);
                const char* bufc = reinterpret_cast<const char*>(OUT) + buf_off;
#pragma nounroll
                for (int j = 0; j < small_trips; ++j) {
                    const int idx = k0 + lane + 64 * j;
                    if (idx < k1) {
                        const f64x2 v = reinterpret_cast<const f64x2*>(bufc)[idx];
                        f64x2* dst = reinterpret_cast<f64x2*>(line0) + idx;
                        if constexpr (OVERWRITE) __builtin_nontemporal_store(v, dst);
                        else { const f64x2 o = *dst; f64x2 r; r.x = o.x + v.x; r.y = o.y + v.y; *dst = r; }
                    }
                }
                // the ends of a run of positions are single doubles: lane 0 the lower, lane 1 the upper one
                const int e = lane == 0 ? (((lo & 1) && lo < L) ? lo : -1) : (((L & 1) && L - 1 >= lo) ? L - 1 : -1);
                if (lane < 2 && e >= 0) {
                    const double v = reinterpret_cast<const double*>(bufc)[e];
                    if constexpr (OVERWRITE) __builtin_nontemporal_store(v, line0 + e); else line0[e] += v;
                }
                if (cout && lane < hi - L) reinterpret_cast<double*>(reinterpret_cast<char*>(OUT) + other_off)[lane] = reinterpret_cast<const double*>(bufc)[L + lane];
            };
            const unsigned out_stride = (unsigned)(accp * 8);
            lds_barrier();  // B0
            tr_start();
            int cin = 0;
            unsigned cur = 0u;                       // byte offset of the buffer the row waves fill during this position
            int4 h_prev = vec(HDR[p_begin & 3]);
            if constexpr (MASKED) { if (!(rfl(h_prev.z) & 1)) tr_barrier(); }
            tr_barrier();
            cur ^= out_stride;
            for (int p = p_begin + 1; p < p_end; ++p) {
                const int4 h_cur = vec(HDR[p & 3]);
                if constexpr (MASKED) { if (!(rfl(h_cur.z) & 1)) tr_barrier(); }   // an incomplete position: the row waves clear their buffer first (see there)
                int cout = (h_cur.x == h_prev.x + h_prev.y && !(ablate_arg & AFFINE_ROWS_NO_CARRY)) ? 1 : 0;
                {   // a position of less than two lines that ends before the first line boundary behind its start stores its own piece (see below)
                    const int head_p = h_prev.w & 15, lo_p = cin ? 0 : head_p, hi_p = head_p + h_prev.y;
                    if ((h_prev.z & 8) && (hi_p & ~15) < lo_p) cout = 0;
                }
                stream_small(h_prev, cur ^ out_stride, cur, cin, cout);
                cin = cout;
                h_prev = h_cur;
                tr_barrier();
                cur ^= out_stride;
            }
            stream_small(h_prev, cur ^ out_stride, cur, cin, 0);
            if (wave == 5) tr_report(2);
            return;
        }
        lds_barrier();  // B0
        tr_start();
        bool carry_in = false;
        int par = 0;
        for (int p = p_begin; p < p_end; ++p, par ^= 1) {
            if constexpr (MASKED) { if (!(rfl(HDR[p & 3].z) & 1)) tr_barrier(); }   // an incomplete position: the row waves clear their buffer first (see there)
            if (p > p_begin) {
                const int4 h_prev = HDR[(p - 1) & 3];
                const int r0_cur = rfl(HDR[p & 3].x);
                bool carry_out = r0_cur == rfl(h_prev.x) + rfl(h_prev.y) && !(ablate_arg & AFFINE_ROWS_NO_CARRY);
                if (carry_out && (rfl(h_prev.z) & 8)) {   // (only a position of less than two lines, flagged by the builder, can be one; the test below is exact)
                    // A position whose rows end before the first line boundary behind their start -- a node without elements (empty rows)
                    // or a single short row -- has nothing to store now, and what it would hand on starts at the line's beginning: below
                    // its own first value lies whatever the staging buffer held (zeros), and the next position would write that over the
                    // end of the PREVIOUS rows in memory, which another position owns.  Such a position stores its own piece of the line
                    // itself (found by scripts/fuzz_gather.py on a box with holes; the structured meshes never have one).
                    const int head_p = rfl(h_prev.w) & 15, lo_p = carry_in ? 0 : head_p, hi_p = head_p + SS * rfl(h_prev.y);
                    if ((hi_p & ~15) < lo_p) carry_out = false;
                }
                stream_out(h_prev, OUT + (size_t)(par ^ 1) * accp, OUT + (size_t)par * accp, carry_in, carry_out);
                carry_in = carry_out;
            }
            tr_barrier();
        }
        stream_out(HDR[(p_end - 1) & 3], OUT + (size_t)(par ^ 1) * accp, OUT + (size_t)par * accp, carry_in, false);
        if (wave == 5) tr_report(2);
        return;
    }



    // ring entry .w: head | extent of the position's rows in doubles << 4 (what an incomplete position clears, below)
    auto with_head = [&](int4 h) { h.w = MASKED ? (head_of(h.x) | ((SS * h.y) << 4)) : (head_of(h.x) | (h.w << 8)); return h; };
    if (wave == 4) {
        // ------------------------------------------------------------------------------------------ loader wave
        // Every global load of the kernel: the element records of the next position's slots (R or M, GW doubles each, written
        // per element by k_affine_records right before this kernel; 16 bytes per lane and round, parked in LDS), the lane
        // table when it changes (positions with identical lane records share a table: the interior of a structured mesh
        // never changes it), the position headers (into the ring, two positions ahead).  Everything is consumed in place a
        // position after it was requested, so the wave waits for a fetch only when memory takes longer than a whole position;
        // the row waves and the store wave never touch vmcnt.
        const int lane = tid - 256;
        { const int pr = (ablate_arg >> (AFFINE_ROWS_PRIO_SHIFT + 2)) & 3; if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); }
        constexpr int NPC = GW / 2;                    // 16-byte pieces per record
        constexpr int ROUNDS = (NPC * 32 + 63) / 64;   // us <= 32 slots
        const int npieces = NPC * T.us;
        auto slot_of = [&](int r) { return min(lane + 64 * r, npieces - 1) / NPC; };
        auto piece_of = [&](int r) { const int i = min(lane + 64 * r, npieces - 1); return i - (i / NPC) * NPC; };
        // (profiling, instrumented instantiation: 256 no element-id fetches -- the record of slot s of position p is taken from element
        // 32 p + s --, 512 only the first of the three record rounds)
        auto load_elem = [&](int p, int r) {
            if (DBG && (ablate & 256)) return (int)(((unsigned)p * 32u + (unsigned)slot_of(r)) & 0x7fffffu);
            return T.elem[(size_t)((unsigned)min(p, npos - 1) * (unsigned)T.us + (unsigned)slot_of(r))];
        };
        // ablate 128 (profiling): every record from the first 4096 (cache-resident): the same instruction stream without the HBM reads
        auto load_piece = [&](int e, int r) {
            if (DBG && (ablate & 512) && r > 0) return f64x2{0.0, 0.0};
            return reinterpret_cast<const f64x2*>(T.rec)[(size_t)(unsigned)((DBG && (ablate & 128)) ? (max(e, 0) & 4095) : max(e, 0)) * NPC + piece_of(r)];
        };
        auto park_piece = [&](int parity, int r, f64x2 v) {
            if (lane + 64 * r < npieces) reinterpret_cast<f64x2*>(JS + ((size_t)parity * T.us + slot_of(r)) * GW)[piece_of(r)] = v;
        };
        auto load_tab = [&](int id, int half) { return reinterpret_cast<const uint4*>(T.lanes)[(size_t)(unsigned)id * 128u + 64u * half + lane]; };
        auto park_tab = [&](int slot, int half, uint4 v) { reinterpret_cast<uint4*>(LT + 256 * slot)[64 * half + lane] = v; };
        // (.z bit 4: the position lies behind this workgroup's range -- the Laplace row loop leaves on it instead of counting positions)
        auto ring_entry = [&](int4 hq, int slot, bool changed, bool beyond) {
            int4 o = with_head(hq);
            o.z = (hq.z & 9) | (slot << 1) | (changed ? 4 : 0) | (beyond ? 16 : 0) | (hq.z & ~0xff);
            return o;
        };
        // prologue: ring entries, lane tables and records of p_begin (and what p_begin + 1 needs), fetches for the next ones
        int4 hq0 = T.hdr[p_begin], hq1 = T.hdr[min(p_begin + 1, npos - 1)];
        int slot_cur = 0;                                              // table slot of position p + 1 while p is current
        int id_prev = hq0.z >> 8;
        {
            const uint4 t0 = load_tab(id_prev, 0), t1 = load_tab(id_prev, 1);
            park_tab(0, 0, t0); park_tab(0, 1, t1);
            const bool ch1 = (hq1.z >> 8) != id_prev;
            if (ch1) { const uint4 u0 = load_tab(hq1.z >> 8, 0), u1 = load_tab(hq1.z >> 8, 1); park_tab(1, 0, u0); park_tab(1, 1, u1); slot_cur = 1; }
            if (lane == 0) { HDR[p_begin & 3] = ring_entry(hq0, 0, true, false); HDR[(p_begin + 1) & 3] = ring_entry(hq1, slot_cur, ch1, p_begin + 1 >= p_end); }
            id_prev = hq1.z >> 8;
        }
        // Requests run DEPTH positions ahead of their use (memory answers in 1 - 2 us while the stores of every workgroup are
        // in flight, a position takes well under one): stage k of the registers below belongs to the positions p with
        // (p - p_begin) mod DEPTH = k.  While p is current, its stage holds the records of p + 1 (parked now), the element ids
        // of p + 1 + DEPTH (their records are requested now) and the header of p + 2 (ring entry now); each is refilled in
        // place with what the stage needs DEPTH positions later.
        f64x2 piece[DEPTH][ROUNDS];
        int e_nxt[DEPTH][ROUNDS];
        int4 h_nxt[DEPTH];
        uint4 tab0 = {0, 0, 0, 0}, tab1 = {0, 0, 0, 0};
        bool tab_pending = false;                                      // tab0 / tab1 hold the lane table of position p + 1
        int slot_pending = 0;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) park_piece(0, r, load_piece(load_elem(p_begin, r), r));
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            h_nxt[k] = T.hdr[min(p_begin + k + 2, npos - 1)];
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int e1 = load_elem(p_begin + k + 1, r);
                e_nxt[k][r] = load_elem(p_begin + k + 1 + DEPTH, r);
                piece[k][r] = load_piece(e1, r);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see the row waves
        lds_barrier();  // B0
        tr_start();
        int par = 0;
        // "position p / p + 1 is incomplete" (it takes one barrier more): scalar, shifted along with the headers
        bool inc_cur = MASKED && !(__builtin_amdgcn_readfirstlane(hq0.z) & 1), inc_nxt = MASKED && !(__builtin_amdgcn_readfirstlane(hq1.z) & 1);
        (void)inc_cur; (void)inc_nxt;
        // one position of the loader (a macro, not a lambda: the stages must stay in registers)
#define AFFINE_LOADER_STEP(k, p)                                                                                              \
        {                                                                                                                     \
            if constexpr (MASKED) {                                                                                           \
                if (inc_cur) tr_barrier();              /* incomplete position: see the row waves */                           \
                inc_cur = inc_nxt;                                                                                            \
                inc_nxt = !(__builtin_amdgcn_readfirstlane(h_nxt[k].z) & 1);   /* (still the header of p + 2 here) */        \
            }                                                                                                                 \
            /* in place: what was requested DEPTH positions ago goes to LDS, the next requests go out */                     \
            unsigned long long tq0 = 0;                                                                                       \
            if (DBG && a.trace) tq0 = __builtin_readcyclecounter();                                                           \
            if (!(DBG && (ablate & 4))) {                                                                                     \
                _Pragma("unroll") for (int r = 0; r < ROUNDS; ++r) {                                                          \
                    park_piece(par ^ 1, r, piece[k][r]);              /* records of p + 1 */                                  \
                    piece[k][r] = load_piece(e_nxt[k][r], r);         /* records of p + 1 + DEPTH */                          \
                    e_nxt[k][r] = load_elem((p) + 1 + 2 * DEPTH, r);                                                          \
                }                                                                                                             \
            }                                                                                                                 \
            if (DBG && a.trace) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tr_seg += __builtin_readcyclecounter() - tq0; } \
            /* lane table of p + 1, requested a position ago, into its slot (nobody reads that slot during this position) */ \
            if (tab_pending) { park_tab(slot_pending, 0, tab0); park_tab(slot_pending, 1, tab1); tab_pending = false; }       \
            /* position p + 2: ring entry, and the request for its lane table if it differs from that of p + 1 (parked      \
               during p + 1, read at the top of p + 2) */                                                                     \
            const int id2 = __builtin_amdgcn_readfirstlane(h_nxt[k].z) >> 8;                                                  \
            const bool ch2 = id2 != id_prev;                                                                                  \
            const int slot2 = ch2 ? (slot_cur ^ 1) : slot_cur;                                                                \
            if (lane == 0) HDR[((p) + 2) & 3] = ring_entry(h_nxt[k], slot2, ch2, (p) + 2 >= p_end);                           \
            if (ch2) { tab0 = load_tab(id2, 0); tab1 = load_tab(id2, 1); tab_pending = true; slot_pending = slot2; }          \
            slot_cur = slot2;                                                                                                 \
            id_prev = id2;                                                                                                    \
            h_nxt[k] = T.hdr[min((p) + 2 + DEPTH, npos - 1)];                                                                 \
            tr_barrier();                                                                                                     \
            par ^= 1;                                                                                                         \
        }
        // whole groups of DEPTH positions, then the rest: a loop that can be left between two stages makes the compiler wait
        // for the youngest requests at its top (the exits share the latch)
        int p0 = p_begin;
        for (; p0 + DEPTH <= p_end; p0 += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) AFFINE_LOADER_STEP(k, p0 + k)
        }
#pragma unroll
        for (int k = 0; k < DEPTH - 1; ++k)
            if (p0 + k < p_end) AFFINE_LOADER_STEP(k, p0 + k)
#undef AFFINE_LOADER_STEP
        tr_report(1);
        return;
    }

    // ---------------------------------------------------------------------------------------------- row waves
    // No global memory traffic at all: lane records, element records and headers come through LDS.
    lds_barrier();  // B0
    tr_start();
    const unsigned hdr_addr = (unsigned)(unsigned long long)HDR + 8u;   // .z (flags | slot << 1 | changed << 2 | id << 8), .w (head)
    uint2 lane_cur = {0u, 0u};
    bool zero_lane = false;        // this lane belongs to a block WITHOUT a term (k_build_affine_rows: element masks): it stores zeros
    bool any_zero_lane = false;    // ... some lane of this wavefront does (scalar)
    f64x2 gq0[3] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}}, gq1[3] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};   // Laplace: Ghat of the lane's terms
    int par = 0;
    if constexpr (LAP && !DBG) {
        // Laplace: two nested loops -- the outer one runs once per lane table, the inner one over the positions that share it -- so that
        // everything derived from the lane record (the reference blocks of the lane's two terms, the record offsets, the place of the
        // block in the staged rows) lives in fixed registers: as ONE loop with a reload under `if (table changed)` the compiler carried
        // thirteen register pairs through two copies per position, a quarter of this wave's ~100 instructions (round 4, C2).
        // The scalar unit is what this loop is short of (profiles/r06_c2_lds_bound.txt: sixteen more scalar instructions per position cost 12 %,
        // sixteen vector ones 1 %): the buffer offsets and the ring offset move by one scalar instruction each, the DPP sums add a selected zero
        // and the lanes without a block store to a dump slot instead of branching, and the loop ends on ONE bit of the header (table changed | behind
        // the range, the second set by the loader) instead of counting positions.
        int z = 0, head = 0;
        unsigned hoff = 16u * (unsigned)(p_begin & 3);
        const unsigned js_stride = (unsigned)(T.us * GW * 8), out_stride = (unsigned)(accp * 8);
        unsigned js_cur = 0u, out_cur = 0u;
        auto read_hdr = [&]() {
            int zw[2];
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*reinterpret_cast<long long*>(zw)) : "v"(hdr_addr + hoff) : "memory");
            z = __builtin_amdgcn_readfirstlane(zw[0]);
            head = zw[1] & 15;
            if (MASKED && !(z & 17)) {   // an incomplete position (of this range): its row lanes clear the extent first (see the general loop below)
                const int ext = __builtin_amdgcn_readfirstlane(zw[1]) >> 4;
                double* buf = reinterpret_cast<double*>(reinterpret_cast<char*>(OUT) + out_cur);
                if (!(ablate_arg & AFFINE_ROWS_NO_CLEAR)) {
                    const int lo = head, hi = head + ext, k0 = (lo + 1) >> 1, k1 = hi >> 1;
                    const f64x2 z2 = {0.0, 0.0};
                    for (int k = k0 + tid; k < k1; k += 256) reinterpret_cast<f64x2*>(buf)[k] = z2;
                    if (tid == 0 && (lo & 1) && lo < hi) buf[lo] = 0.0;
                    if (tid == 1 && (hi & 1) && hi - 1 >= lo) buf[hi - 1] = 0.0;
                }
                tr_barrier();
            }
        };
        read_hdr();
        const unsigned out_base = (unsigned)(unsigned long long)OUT, dump_addr = (unsigned)(unsigned long long)DUMP + 8u * (unsigned)(tid & 7);
        while (!(z & 16)) {
            const uint2 lc = LT[256 * ((z >> 1) & 1) + tid];
            const unsigned x = lc.x, y = lc.y;
            bool zl = false, anyz = false;
            if constexpr (MASKED) {
                zl = ((x >> 5) & 127u) == AR_ZERO_G && ((x >> 17) & 127u) == AR_ZERO_G && ((x >> 28) & 1u);
                anyz = __builtin_amdgcn_ballot_w64(zl) != 0ull;
            }
            const f64x2* q0 = reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(GH) + ((x >> 5) & 127u) * (GW * 8));
            const f64x2* q1 = reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(GH) + ((x >> 17) & 127u) * (GW * 8));
            const f64x2 a0 = q0[0], a1 = q0[1], a2 = q0[2], b0 = q1[0], b1 = q1[1], b2 = q1[2];
            const unsigned oR0 = (x & 31u) * (GW * 8), oR1 = ((x >> 12) & 31u) * (GW * 8);
            const int grp = (int)((x >> 24) & 3u);
            const bool stores = ((x >> 28) & 1u) != 0u;
            const unsigned yoff = out_base + (y & 0xffffu);
            for (;;) {
                const char* js = reinterpret_cast<const char*>(JS) + js_cur;
                const f64x2* m0 = reinterpret_cast<const f64x2*>(js + oR0);
                const f64x2* m1 = reinterpret_cast<const f64x2*>(js + oR1);
                const f64x2 u0 = m0[0], u1 = m0[1], u2 = m0[2], w0 = m1[0], w1 = m1[1], w2 = m1[2];
                double sm = a0.x * u0.x;
                sm = fma(a0.y, u0.y, sm); sm = fma(a1.x, u1.x, sm); sm = fma(a1.y, u1.y, sm); sm = fma(a2.x, u2.x, sm); sm = fma(a2.y, u2.y, sm);
                sm = fma(b0.x, w0.x, sm); sm = fma(b0.y, w0.y, sm); sm = fma(b1.x, w1.x, sm); sm = fma(b1.y, w1.y, sm); sm = fma(b2.x, w2.x, sm);
                sm = fma(b2.y, w2.y, sm);
                { const double t = dpp_quad_full<0xB1>(sm); sm += (grp >= 1) ? t : 0.0; }
                { const double t = dpp_quad_full<0x4E>(sm); sm += (grp >= 2) ? t : 0.0; }
                if constexpr (MASKED) { if (anyz && zl) sm = 0.0; }
                {
                    const unsigned dst = stores ? yoff + out_cur + 8u * (unsigned)head : dump_addr;
                    asm volatile("ds_write_b64 %0, %1" : : "v"(dst), "v"(sm) : "memory");
                }
                tr_barrier();
                hoff = (hoff + 16u) & 48u;
                js_cur ^= js_stride;
                out_cur ^= out_stride;
                read_hdr();
                if (z & 20) break;   // the lane table changes with this position, or the range is done
            }
        }
        if (wave == 0) tr_report(0);
        return;
    }
    for (int p = p_begin; p < p_end; ++p, par ^= 1) {
        int zw[2];
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*reinterpret_cast<long long*>(zw)) : "v"(hdr_addr + 16u * (unsigned)(p & 3)) : "memory");
        const int z = __builtin_amdgcn_readfirstlane(zw[0]);
        const int head = zw[1] & 15;
        if (MASKED && !(z & 1)) {
            // A position with a block that has no owner lane (element masks: the interface planes of a partition, a per cent of the
            // positions) needs that block to read as zero, whatever the buffer held before -- the position two steps back, complete or
            // not, with rows of another extent.  Its own row lanes clear the extent first (a few stores each; the store wave, the
            // critical path, is left alone: with the clearing there the workgroups that own an interface plane finished 20 % late and
            // the masked sweep took 5.9 instead of 4.95 ms), then every role of the workgroup meets at one extra barrier.  The carried
            // piece below `head` belongs to the store wave.  (Until round 3 only incomplete positions cleared, and only behind
            // themselves: with several positions per workgroup an incomplete one inherited the values of a complete one.)
            const int ext = __builtin_amdgcn_readfirstlane(zw[1]) >> 4;
            double* buf = OUT + (size_t)par * accp;
            if (!(ablate_arg & AFFINE_ROWS_NO_CLEAR)) {   // (timing experiments: the barrier without the stores)
                const int lo = head, hi = head + ext, k0 = (lo + 1) >> 1, k1 = hi >> 1;   // whole 16-byte pieces [k0, k1), single doubles at odd ends
                const f64x2 z2 = {0.0, 0.0};
                for (int k = k0 + tid; k < k1; k += 256) reinterpret_cast<f64x2*>(buf)[k] = z2;
                if (tid == 0 && (lo & 1) && lo < hi) buf[lo] = 0.0;
                if (tid == 1 && (hi & 1) && hi - 1 >= lo) buf[hi - 1] = 0.0;
            }
            tr_barrier();
        }
        if (z & 4) {   // the lane table changed with this position
            lane_cur = LT[256 * ((z >> 1) & 1) + tid];
            if constexpr (MASKED) {
                zero_lane = ((lane_cur.x >> 5) & 127u) == AR_ZERO_G && ((lane_cur.x >> 17) & 127u) == AR_ZERO_G && ((lane_cur.x >> 28) & 1u);
                any_zero_lane = __builtin_amdgcn_ballot_w64(zero_lane) != 0ull;
            }
            if constexpr (LAP) {  // ... and with it the reference blocks of this lane's two terms: kept in registers
                const f64x2* q0 = reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(GH) + ((lane_cur.x >> 5) & 127u) * (GW * 8));
                const f64x2* q1 = reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(GH) + ((lane_cur.x >> 17) & 127u) * (GW * 8));
#pragma unroll
                for (int h = 0; h < 3; ++h) { gq0[h] = q0[h]; gq1[h] = q1[h]; }
            }
        }

        const unsigned x = lane_cur.x, y = lane_cur.y;
        const char* js = reinterpret_cast<const char*>(JS + (size_t)par * T.us * GW);
        const char* gh = reinterpret_cast<const char*>(GH);
        const unsigned oR0 = (x & 31u) * (GW * 8), oG0 = ((x >> 5) & 127u) * (GW * 8);
        const unsigned oR1 = ((x >> 12) & 31u) * (GW * 8), oG1 = ((x >> 17) & 127u) * (GW * 8);
        const int grp = (int)((x >> 24) & 3u);
        // (a zero lane stores zeros whatever slot 0 holds: the record of an empty slot may be anything, and 0 x NaN is not 0)
        char* out_par = reinterpret_cast<char*>(OUT + (size_t)par * accp);
        if constexpr (LAP) {
            const f64x2* m0 = reinterpret_cast<const f64x2*>(js + oR0);
            const f64x2* m1 = reinterpret_cast<const f64x2*>(js + oR1);
            double s = 0.0;
            if (!(DBG && (ablate & 2))) {
#pragma unroll
                for (int h = 0; h < 3; ++h) { const f64x2 m = m0[h], g = gq0[h]; s = fma(g.x, m.x, s); s = fma(g.y, m.y, s); }
#pragma unroll
                for (int h = 0; h < 3; ++h) { const f64x2 m = m1[h], g = gq1[h]; s = fma(g.x, m.x, s); s = fma(g.y, m.y, s); }
            }
            if (grp >= 1) s += dpp_quad_full<0xB1>(s);
            if (grp >= 2) s += dpp_quad_full<0x4E>(s);
            if constexpr (MASKED) { if (any_zero_lane && zero_lane) s = 0.0; }
            if ((x >> 28) & 1u) *reinterpret_cast<double*>(out_par + 8 * head + (y & 0xffffu)) = s;
        } else {
            double H[3][3];
            if (!(DBG && (ablate & 2))) {
                auto load33 = [&](const char* p_, double (&M)[3][3]) {
                    const f64x2* q = reinterpret_cast<const f64x2*>(p_);
                    const f64x2 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4];
                    M[0][0] = v0.x; M[0][1] = v0.y; M[0][2] = v1.x; M[1][0] = v1.y; M[1][1] = v2.x; M[1][2] = v2.y;
                    M[2][0] = v3.x; M[2][1] = v3.y; M[2][2] = v4.x;
                };
                // one term after the other (the second term's operands are fetched while the first is multiplied): H stays
                // one chain of six products per entry
                auto term = [&](const char* pr, const char* pg, bool first) {
                    double R[3][3], Gm[3][3], Tm[3][3];
                    load33(pr, R);
                    load33(pg, Gm);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int s_ = 0; s_ < 3; ++s_) Tm[c][s_] = fma(Gm[c][2], R[2][s_], fma(Gm[c][1], R[1][s_], Gm[c][0] * R[0][s_]));
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int s_ = 0; s_ < 3; ++s_) {
                            double h = first ? R[0][i] * Tm[0][s_] : fma(R[0][i], Tm[0][s_], H[i][s_]);
                            h = fma(R[1][i], Tm[1][s_], h);
                            H[i][s_] = fma(R[2][i], Tm[2][s_], h);
                        }
                };
                term(js + oR0, gh + oG0, true);
                term(js + oR1, gh + oG1, false);
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) H[i][s_] = (double)(x + 3 * i + s_);
            }
            if (grp >= 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) H[i][s_] += dpp_quad_full<0xB1>(H[i][s_]);
            }
            if (grp >= 2) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) H[i][s_] += dpp_quad_full<0x4E>(H[i][s_]);
            }
            if (MASKED && any_zero_lane) {   // (scalar branch)
                if (zero_lane) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int s_ = 0; s_ < 3; ++s_) H[i][s_] = 0.0;
                }
            }
            if ((x >> 28) & 1u) {
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
                const unsigned rs = y >> 16;
                char* stage = out_par + 8 * head + (y & 0xffffu);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double* row = reinterpret_cast<double*>(stage + i * rs);
                    row[0] = v[i][0]; row[1] = v[i][1]; row[2] = v[i][2];
                }
            }
        }
        tr_barrier();
    }
    if (wave == 0) tr_report(0);
}

// ------------------------------------------------------------------------------------------------ element records
// One thread per element, once per assembly, right before k_affine_rows: the record of an affine element from four of its
// vertices.  The edges from node 0 to nodes 1, 3, 4 are twice the columns of J (hexahedron.rs:49-58: nodes (---), (+--),
// (-+-), (--+); exact for a parallelepiped; elliptic.rs:398-404 evaluates the same Jacobian at every quadrature point).
// LinearElastic: R = sqrt(|det J|) J^-1 = sign(det J) rsqrt(|det J|) adj(J), nine doubles + pad; Laplace: M = R R^T, six.
// det J == 0 exactly is the reference's "Singular element Jacobian" (try_inverse fails only then): reported, record zero.
template <int OP>
__global__ void __launch_bounds__(256) k_affine_records(const double* verts, const int* conn, const unsigned char* elem_aff,
                                                        const unsigned char* active, long long e_first, long long E, double* rec, DevStatus* status) {
    // OP == FH_MASS_SCALAR (the mass matrix of affine elements, mass.rs:131-286: M_ab = |det J| sum_q w rho phi_a phi_b): the Laplace
    // layout with |det J| in the first place -- k_affine_rows<FH_LAPLACE> multiplies it with the reference block (sum_q w rho phi_a phi_b, 0 ...)
    constexpr bool MASS = (OP == FH_MASS_SCALAR);
    constexpr bool LAP = (OP == FH_LAPLACE) || MASS;
    constexpr int GW = LAP ? AFFINE_ROWS_GW_LAP : AFFINE_ROWS_GW_LE, NPC = GW / 2;
    // the records of the workgroup's 256 consecutive elements are staged in LDS and leave as consecutive 16-byte pieces: a thread's
    // own record is 80 (48) bytes, so stores straight from the registers put every lane on a line of its own
    __shared__ f64x2 stage[256 * NPC];
    __shared__ unsigned char ok[256];
    const long long e0 = e_first + (long long)blockIdx.x * 256;
    const long long e = e0 + threadIdx.x;
    const bool mine = e < E && elem_aff[e];
    ok[threadIdx.x] = mine ? 1 : 0;
    if (mine) {
        const int4 c0 = reinterpret_cast<const int4*>(conn)[2 * e], c1 = reinterpret_cast<const int4*>(conn)[2 * e + 1];
        const int vi[4] = {c0.x, c0.y, c0.w, c1.x};
        double X[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) X[k][c] = verts[(size_t)vi[k] * 3 + c];
        double J[3][3], R[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) J[i][k] = 0.5 * (X[k + 1][i] - X[0][i]);
        const double detJ = det_small<3>(J);
        if constexpr (MASS) {   // (no inverse, no singular report: a degenerate element contributes nothing)
            f64x2* om = stage + threadIdx.x * NPC;
            f64x2 v0; v0.x = fabs(detJ); v0.y = 0.0;
            const f64x2 z2 = {0.0, 0.0};
            om[0] = v0; om[1] = z2; om[2] = z2;
        } else
        if (detJ == 0.0) {
            if (!active || active[e]) report_singular(status, e);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) R[i][j] = 0.0;
        } else {
            adj_scaled(J, copysign(rsqrt_newton(fabs(detJ)), detJ), R);
        }
        f64x2* o = stage + threadIdx.x * NPC;
        if constexpr (MASS) {
            (void)o;
        } else if constexpr (LAP) {
            double M[6];
            int k = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int d = c; d < 3; ++d, ++k) M[k] = R[c][0] * R[d][0] + R[c][1] * R[d][1] + R[c][2] * R[d][2];
#pragma unroll
            for (int h = 0; h < 3; ++h) { f64x2 v; v.x = M[2 * h]; v.y = M[2 * h + 1]; o[h] = v; }
        } else {
            const double r9[10] = {R[0][0], R[0][1], R[0][2], R[1][0], R[1][1], R[1][2], R[2][0], R[2][1], R[2][2], 0.0};
#pragma unroll
            for (int h = 0; h < 5; ++h) { f64x2 v; v.x = r9[2 * h]; v.y = r9[2 * h + 1]; o[h] = v; }
        }
    }
    __syncthreads();
    f64x2* out = reinterpret_cast<f64x2*>(rec + (size_t)e0 * GW);
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int idx = threadIdx.x + 256 * k;
        if (ok[idx / NPC]) out[idx] = stage[idx];   // records of elements that are not affine stay untouched
    }
}

hipError_t affine_records_launch(int op, hipStream_t stream, const double* verts, const int* conn, const unsigned char* elem_aff,
                                 const unsigned char* active, long long e_first, long long e_end, double* rec, DevStatus* status) {
    if (e_end <= e_first) return hipSuccess;
    const dim3 grid((unsigned)((e_end - e_first + 255) / 256));
    if (op == FH_LAPLACE) hipLaunchKernelGGL(k_affine_records<FH_LAPLACE>, grid, dim3(256), 0, stream, verts, conn, elem_aff, active, e_first, e_end, rec, status);
    else if (op == FH_MASS_SCALAR) hipLaunchKernelGGL(k_affine_records<FH_MASS_SCALAR>, grid, dim3(256), 0, stream, verts, conn, elem_aff, active, e_first, e_end, rec, status);
    else hipLaunchKernelGGL(k_affine_records<FH_LINEAR_ELASTIC>, grid, dim3(256), 0, stream, verts, conn, elem_aff, active, e_first, e_end, rec, status);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ table builder
// One workgroup (one wave) per position.  Input: the position record of k_build_pipe_tables (header, entries
// slot << 16 | local a << 8 | block-local node, per (entry, local node) the column slot in the owner's row, relative row
// offsets).  Terms are grouped by output block (node, column slot), ordered by element id, and dealt two per lane.
__global__ void __launch_bounds__(64) k_build_affine_rows(const int* p_rec, int rw_old, int us, int ms, int nbs, int npos, int S,
                                                          const unsigned* ncols, const int* p_conn, int cs, const int* p_elem,
                                                          int4* hdr_out, uint2* lanes, int* status,
                                                          unsigned long long* hash_out, const int mirror, const int* plist,
                                                          unsigned long long* hash2_out, const int ksh) {
    // plist == null: one workgroup per position: header, two 64-bit hashes of the 256 records, and -- when `lanes` is given -- the
    // records themselves at lanes[p].  plist != null (second pass of the hash-only build): workgroup t forms the records of position
    // plist[t] once more and writes them to lanes[t] (the compact table), nothing else.
    // keys (block-local node, column slot) = il << ksh | pos with 2^ksh >= the longest node row of the pattern (round 5: 32 on a hexahedral
    // mesh instead of a fixed 128 -- a quarter of the key loops' trips and 7 instead of 22 KB of LDS, i.e. three times the workgroups per CU)
    constexpr int N = 8, TMAX = 8;
    const int KS = 1 << ksh, NKEY = 8 * KS;
    extern __shared__ __attribute__((aligned(16))) char build_smem[];
    int* cnt = reinterpret_cast<int*>(build_smem);                                   // [NKEY]
    unsigned* lw0 = reinterpret_cast<unsigned*>(cnt + NKEY);                         // [256]
    unsigned* lw1 = lw0 + 256;                                                       // [256]
    unsigned short* bucket = reinterpret_cast<unsigned short*>(lw1 + 256);           // [NKEY * TMAX]
    const int p = plist ? plist[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
    const int* rec = p_rec + (size_t)p * rw_old;
    const GatherHdr h = *reinterpret_cast<const GatherHdr*>(rec);
    const unsigned* ent = reinterpret_cast<const unsigned*>(rec + 8 + us / 4);
    const unsigned char* posb = reinterpret_cast<const unsigned char*>(rec + 8 + us / 4 + ms);
    const int* noff_old = rec + 8 + us / 4 + ms + ms * N / 4;
    const int* el = p_elem + (size_t)p * us;
    for (int i = lane; i < NKEY; i += 64) cnt[i] = 0;
    const unsigned idle = (AR_ZERO_G << 5) | (AR_ZERO_G << 17);
    for (int i = lane; i < 256; i += 64) { lw0[i] = idle; lw1[i] = 0u; }
    __syncthreads();
    bool bad = false;
    for (int idx = lane; idx < h.m * N; idx += 64) {
        const int t = idx / N, j = idx % N;
        const unsigned e = ent[t];
        const unsigned slot = e >> 16, a_loc = (e >> 8) & 0xffu, il = e & 0xffu, pos = posb[t * N + j];
        if (il >= 8u || pos >= (unsigned)KS || slot >= 32u || a_loc >= 8u) { bad = true; continue; }
        const int key = (int)((il << ksh) + pos);
        const int s_ = atomicAdd(&cnt[key], 1);
        if (s_ < TMAX) bucket[key * TMAX + s_] = (unsigned short)(slot | (a_loc << 8) | ((unsigned)j << 11));
        else bad = true;
    }
    __syncthreads();
    // fixed order of the terms of a block (the atomics above hand out positions in arbitrary order): an element meets a
    // block once, so the element id alone orders them -- the same order for the owners of (I, J) and (J, I)
    for (int key = lane; key < NKEY; key += 64) {
        const int Tn = min(cnt[key], TMAX);
        unsigned short* b = bucket + key * TMAX;
        for (int i = 1; i < Tn; ++i) {
            const unsigned short v = b[i];
            const int kv = el[v & 255u];
            int k = i - 1;
            while (k >= 0 && el[b[k] & 255u] > kv) { b[k + 1] = b[k]; --k; }
            b[k + 1] = v;
        }
    }
    __syncthreads();
    // classes: 5..8 terms -> 4 lanes, 3..4 -> 2 lanes, 1..2 -> one lane; a block of these rows WITHOUT a term (element masks: no active
    // element joins the two nodes) gets a lane of its own that stores zeros, when the lanes suffice -- the position is then complete and
    // never needs its staging buffer cleared (the interface planes of a slab partition: ~140 owner lanes + 63 such blocks)
    auto is_block = [&](int key) { const int il = key >> ksh, pos = key & (KS - 1); return il < h.nb && pos < noff_old[il + 1] - noff_old[il]; };
    // mirror (k_hex8_rows): when both nodes of a block (I, J) are owned by this position, only the owner of the smaller node keeps
    // lanes for it; they store the block to (I, J) and its transpose to (J, I).  twin_of(key) = block-local index of J when the block
    // has such a partner (-1: none); skipped(key): this is the partner's copy
    auto twin_of = [&](int key) {
        if (!mirror || !is_block(key)) return -1;
        const int il = key >> ksh, pos = key & (KS - 1);
        const long long jl = (long long)ncols[(size_t)h.r0 + noff_old[il] + pos] - (long long)h.i0;
        return (jl >= 0 && jl < h.nb && jl != il) ? (int)jl : -1;
    };
    auto skipped = [&](int key) { const int jl = twin_of(key); return jl >= 0 && jl < (key >> ksh); };
    for (int key = lane; key < NKEY; key += 64)
        if (skipped(key)) cnt[key] = -1;      // no lane, and not a block without a term either
    __syncthreads();
    int n4 = 0, n2 = 0, n1 = 0, n0 = 0, nskip = 0;
    for (int base = 0; base < NKEY; base += 64) {
        const int Tn = min(cnt[base + lane], TMAX);
        n4 += __popcll(__ballot(Tn >= 5));
        n2 += __popcll(__ballot(Tn == 3 || Tn == 4));
        n1 += __popcll(__ballot(Tn == 1 || Tn == 2));
        n0 += __popcll(__ballot(Tn == 0 && is_block(base + lane)));
        nskip += __popcll(__ballot(Tn < 0));
    }
    const int base4 = 0, base2 = 4 * n4, base1 = base2 + 2 * n2, base0 = base1 + n1;
    if (base1 + n1 > 256) bad = true;
    const bool zero_lanes = n0 > 0 && base0 + n0 <= 256;
    if ((size_t)8 * S * S * (size_t)h.nrow >= 65536u) bad = true;
    if (mirror && (S * S * h.nrow >= 8192 || h.nb > 8)) bad = true;   // the two 13-bit offsets / 3-bit nodes of the hex8 record
    if (__ballot(bad)) {
        if (plist) return;   // (cannot happen: the first pass reported it)
        if (lane == 0) {
            atomicOr(status, 1);
            hdr_out[p] = make_int4(h.r0, h.nrow, 0, h.U);
        }
        if (lanes) for (int i = lane; i < 256; i += 64) lanes[(size_t)p * 256 + i] = make_uint2(idle, 0u);
        if (lane == 0) { hash_out[p] = 0ull; if (hash2_out) hash2_out[p] = 0ull; }
        return;
    }
    int r4 = 0, r2 = 0, r1 = 0, r0 = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = 0; base < NKEY; base += 64) {
        const int key = base + lane;
        const int Tn = min(cnt[key], TMAX);
        const unsigned il = (unsigned)key >> ksh, pos = (unsigned)key & (unsigned)(KS - 1);
        const unsigned short* b = bucket + key * TMAX;
        const unsigned long long m4 = __ballot(Tn >= 5), m2 = __ballot(Tn == 3 || Tn == 4), m1 = __ballot(Tn == 1 || Tn == 2);
        const bool zb = zero_lanes && Tn == 0 && is_block(key);
        const unsigned long long m0 = __ballot(zb);
        if (zb) {   // no term: slot 0 with the block of zeros twice, the store flag, the block's place in the staged rows
            const int rb = noff_old[il], cnt_row = noff_old[il + 1] - rb;
            const int Lidx = base0 + r0 + __popcll(m0 & below);
            unsigned xz = (AR_ZERO_G << 5) | (AR_ZERO_G << 17) | (1u << 28);
            unsigned yz = (unsigned)(8 * (S * S * rb + S * (int)pos)) | ((unsigned)(8 * S * cnt_row) << 16);
            if (mirror) {   // hex8 record (see below); the twin of a block without a term is a block of zeros as well
                yz = (unsigned)(S * S * rb + S * (int)pos) | (il << 13);
                const int jl = twin_of(key);
                if (jl >= 0) {
                    const int rbJ = noff_old[jl], cntJ = noff_old[jl + 1] - rbJ;
                    const unsigned Iz = (unsigned)h.i0 + il;
                    int posJ = 0;
                    while (posJ < cntJ && ncols[(size_t)h.r0 + rbJ + posJ] != Iz) ++posJ;
                    yz |= ((unsigned)(S * S * rbJ + S * posJ) << 16) | ((unsigned)jl << 29);
                    xz |= 1u << 29;
                }
            }
            lw0[Lidx] = xz;
            lw1[Lidx] = yz;
        }
        r0 += __popcll(m0);
        if (Tn >= 1) {
            const int rb = noff_old[il], cnt_row = noff_old[il + 1] - rb;
            const unsigned I = (unsigned)h.i0 + il, J = ncols[(size_t)h.r0 + rb + pos];
            const unsigned trf = I > J ? 1u : 0u, dgf = I == J ? 1u : 0u;
            unsigned yw = (unsigned)(8 * (S * S * rb + S * (int)pos)) | ((unsigned)(8 * S * cnt_row) << 16);
            unsigned twin_bit = 0u;
            if (mirror) {
                // hex8 record: y = offset (doubles) | node << 13 | twin offset << 16 | twin node << 29; the row strides come from the
                // position record (rows per node); x bit 29: the block has a twin
                yw = (unsigned)(S * S * rb + S * (int)pos) | (il << 13);
                const int jl = twin_of(key);
                if (jl >= 0) {
                    const int rbJ = noff_old[jl], cntJ = noff_old[jl + 1] - rbJ;
                    int posJ = 0;
                    while (posJ < cntJ && ncols[(size_t)h.r0 + rbJ + posJ] != I) ++posJ;   // (I is in J's row: the pattern is symmetric)
                    yw |= ((unsigned)(S * S * rbJ + S * posJ) << 16) | ((unsigned)jl << 29);
                    twin_bit = 1u << 29;
                }
            }
            auto lane_words = [&](int first, int Lidx, unsigned grp, bool leader) {
                unsigned xw = (grp << 24) | (trf << 26) | (dgf << 27) | (leader ? ((1u << 28) | twin_bit) : 0u);
                for (int t = 0; t < 2; ++t) {
                    unsigned slot = b[0] & 255u, gidx = AR_ZERO_G;
                    if (first + t < Tn) {
                        const unsigned v = b[first + t], a_loc = (v >> 8) & 7u, j_loc = (v >> 11) & 7u;
                        slot = v & 255u;
                        gidx = trf ? (j_loc * 8u + a_loc) : (a_loc * 8u + j_loc);  // the block of the pair's smaller node
                    }
                    xw |= (slot | (gidx << 5)) << (12 * t);
                }
                lw0[Lidx] = xw;
                lw1[Lidx] = yw;
            };
            if (Tn >= 5) {
                const int L0 = base4 + 4 * (r4 + __popcll(m4 & below));
                for (int g = 0; g < 4; ++g) lane_words(2 * g, L0 + g, 2u, g == 0);
            } else if (Tn >= 3) {
                const int L0 = base2 + 2 * (r2 + __popcll(m2 & below));
                for (int g = 0; g < 2; ++g) lane_words(2 * g, L0 + g, 1u, g == 0);
            } else {
                lane_words(0, base1 + r1 + __popcll(m1 & below), 0u, true);
            }
        }
        r4 += __popcll(m4); r2 += __popcll(m2); r1 += __popcll(m1);
    }
    __syncthreads();
    if (plist) {   // second pass: the records of this table's first position into the compact table
        for (int i = lane; i < 256; i += 64) lanes[(size_t)blockIdx.x * 256 + i] = make_uint2(lw0[i], lw1[i]);
        return;
    }
    // two independent 64-bit hashes of the 256 records, position in the table included (positions with equal tables are merged)
    unsigned long long hsum = 0ull, hsum2 = 0ull;
    for (int i = lane; i < 256; i += 64) {
        if (lanes) lanes[(size_t)p * 256 + i] = make_uint2(lw0[i], lw1[i]);
        const unsigned long long w = (unsigned long long)lw1[i] << 32 | lw0[i];
        unsigned long long z = w + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        hsum += z ^ (z >> 31);
        unsigned long long y = (w ^ 0xD6E8FEB86659FD93ull) * (2ull * (unsigned long long)i + 0xC2B2AE3D27D4EB4Full);
        y = (y ^ (y >> 32)) * 0xFF51AFD7ED558CCDull;
        y = (y ^ (y >> 29)) * 0xC4CEB9FE1A85EC53ull;
        hsum2 += y ^ (y >> 32);
    }
    for (int o = 32; o > 0; o >>= 1) { hsum += __shfl_xor(hsum, o); hsum2 += __shfl_xor(hsum2, o); }
    if (lane == 0) { hash_out[p] = hsum; if (hash2_out) hash2_out[p] = hsum2; }
    // every (node, column) block of these rows has an owner lane: the store wave need not clear the staged rows
    // flags: bit 0 every block of these rows has a lane; bit 3 the rows are shorter than two cache lines (the store wave's carry test)
    if (lane == 0) hdr_out[p] = make_int4(h.r0, h.nrow, ((n4 + n2 + n1 + nskip + (zero_lanes ? n0 : 0) == h.nrow) ? 1 : 0) | ((S * S * h.nrow < 32) ? 8 : 0), h.U);
}

// table id of every position into its header (flags | id << 8), and the first position of every id gathered into the
// compact table; *mismatch is set when a position's records differ from its table's (hash collision)
__global__ void __launch_bounds__(256) k_affine_rows_compact(const uint2* lanes_full, const int* ids, const int* first_pos, int npos,
                                                             int ntab, uint2* lanes_tab, int4* hdr, int* mismatch) {
    const int p = blockIdx.x, t = threadIdx.x;
    if (p < ntab) lanes_tab[(size_t)p * 256 + t] = lanes_full[(size_t)first_pos[p] * 256 + t];
    if (p < npos) {
        const int id = ids[p];
        const uint2 mine = lanes_full[(size_t)p * 256 + t], ref = lanes_full[(size_t)first_pos[id] * 256 + t];
        if (mine.x != ref.x || mine.y != ref.y) *mismatch = 1;
        if (t == 0) {
            if (!(hdr[p].z & 1)) mismatch[1] = 1;   // a block without an owner lane somewhere
            hdr[p].z = (hdr[p].z & 9) | (id << 8);
        }
    }
}

// key shift and LDS bytes of k_build_affine_rows for a pattern whose longest node row has max_row entries
static int build_key_shift(int max_row) { return max_row <= 32 ? 5 : max_row <= 64 ? 6 : 7; }
static size_t build_lds_bytes(int ksh) { const size_t nkey = (size_t)8 << ksh; return nkey * 4 + 2 * 256 * 4 + nkey * 8 * 2; }

hipError_t affine_rows_build(hipStream_t stream, const int* p_rec, int rw_old, int us, int ms, int nbs, int npos, int S,
                             const unsigned* ncols, const int* p_conn, int cs, const int* p_elem, int4* hdr, uint2* lanes,
                             int* status, unsigned long long* hash, int mirror, unsigned long long* hash2, int max_row) {
    if (npos <= 0) return hipSuccess;
    const int ksh = build_key_shift(max_row);
    hipLaunchKernelGGL(k_build_affine_rows, dim3(npos), dim3(64), build_lds_bytes(ksh), stream, p_rec, rw_old, us, ms, nbs, npos, S, ncols, p_conn, cs,
                       p_elem, hdr, lanes, status, hash, mirror, (const int*)nullptr, hash2, ksh);
    return hipGetLastError();
}

// table id of every position into its header; mismatch[1] is set when some position has a block without an owner lane
__global__ void __launch_bounds__(256) k_affine_rows_set_ids(const int* ids, int npos, int4* hdr, int* mismatch) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npos) return;
    const int z = hdr[p].z;
    if (!(z & 1)) mismatch[1] = 1;
    hdr[p].z = (z & 9) | (ids[p] << 8);
}

hipError_t affine_rows_tables(hipStream_t stream, const int* p_rec, int rw_old, int us, int ms, int nbs, int npos, int S, const unsigned* ncols,
                              const int* p_conn, int cs, const int* p_elem, int mirror, const int* ids, const int* first_pos, int ntab,
                              uint2* lanes_tab, int4* hdr, int* mismatch, int max_row) {
    if (npos <= 0 || ntab <= 0) return hipSuccess;
    const int ksh = build_key_shift(max_row);
    hipLaunchKernelGGL(k_build_affine_rows, dim3(ntab), dim3(64), build_lds_bytes(ksh), stream, p_rec, rw_old, us, ms, nbs, npos, S, ncols, p_conn, cs, p_elem,
                       (int4*)nullptr, lanes_tab, (int*)nullptr, (unsigned long long*)nullptr, mirror, first_pos, (unsigned long long*)nullptr, ksh);
    hipLaunchKernelGGL(k_affine_rows_set_ids, dim3((npos + 255) / 256), dim3(256), 0, stream, ids, npos, hdr, mismatch);
    return hipGetLastError();
}

hipError_t affine_rows_compact(hipStream_t stream, const uint2* lanes_full, const int* ids, const int* first_pos, int npos, int ntab,
                               uint2* lanes_tab, int4* hdr, int* mismatch) {
    if (npos <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_affine_rows_compact, dim3(npos > ntab ? npos : ntab), dim3(256), 0, stream, lanes_full, ids, first_pos, npos, ntab,
                       lanes_tab, hdr, mismatch);
    return hipGetLastError();
}

template <int OP, int DEPTH, int NSTORE, bool MASKED>
static auto affine_rows_pick(bool ow, bool dbg) -> void (*)(const KArgs, const AffineRowTables, int) {
    if (dbg) return k_affine_rows<OP, true, true, DEPTH, NSTORE, MASKED>;
    return ow ? k_affine_rows<OP, true, false, DEPTH, NSTORE, MASKED> : k_affine_rows<OP, false, false, DEPTH, NSTORE, MASKED>;
}
template <int OP, bool MASKED>
static auto affine_rows_pick_variant(int depth, bool ow, bool dbg) -> void (*)(const KArgs, const AffineRowTables, int) {
    // depth 3 and beyond: the loader's stages no longer fit the register budget of five waves per SIMD (measured slower); a second store wave
    // (NSTORE = 2) was measured and is no longer instantiated
    return depth <= 1 ? affine_rows_pick<OP, 1, 1, MASKED>(ow, dbg) : affine_rows_pick<OP, 2, 1, MASKED>(ow, dbg);
}

hipError_t affine_rows_launch(int op, int depth, int grid, size_t lds_bytes, hipStream_t stream, const KArgs& a, const AffineRowTables& T, int ablate,
                              bool masked) {
    const bool ow = a.overwrite != 0, dbg = (ablate & 0xffff) != 0;
    void (*kern)(const KArgs, const AffineRowTables, int) =
        masked ? (op == FH_LAPLACE ? affine_rows_pick_variant<FH_LAPLACE, true>(depth, ow, dbg) : affine_rows_pick_variant<FH_LINEAR_ELASTIC, true>(depth, ow, dbg))
               : (op == FH_LAPLACE ? affine_rows_pick_variant<FH_LAPLACE, false>(depth, ow, dbg) : affine_rows_pick_variant<FH_LINEAR_ELASTIC, false>(depth, ow, dbg));
    if (lds_bytes > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(AFFINE_ROWS_THREADS), lds_bytes, stream, a, T, ablate);
    return hipGetLastError();
}

}  // namespace fenris_hip
