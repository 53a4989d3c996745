// Row-owner form of the owner-computes stiffness kernel for Tet4 with a one-point rule (Laplace / LinearElastic, uniform or
// per-element parameters): a lane owns an OUTPUT block (owned node I, column J) and walks a short list of terms
// (slot, local I, local J), so the sums are complete in registers and go to global memory directly -- no ds_add_f64, no row
// accumulators in LDS, no write-out pass.  Same sweep chains, slots, prefetch and phase B as k_gather_pipelined
// (assemble_kernels.hpp); replaces its phase C / finalize / write-out.  Restates elliptic.rs:361-439 + global.rs:133-182 like
// the other kernels.  (The Hex8 form of this idea is k_affine_rows, affine_rows.hip; on general hexahedra the pipelined kernel
// is faster: 64 operand fetches per lane against 48, measured 1.7x.)
#pragma once
#include <hip/hip_runtime.h>

#include "assemble_kernels.hpp"
#include "device_common.hpp"

namespace fenris_hip {

// Tet4 (one-point rule): six terms per lane.  A node of a tetrahedral mesh has ~24 elements and ~15 columns: the diagonal
// block takes 4 lanes (24 terms), everything else one lane.  Lane record: 96 bits (round 6: the kernel is bound by the bytes it moves -- 2.0 GB
// on C3, a third of them these tables -- and no longer by anything it computes, profiles/r06_c3_tables.txt; the record was a uint4 with 16-bit
// terms at a stride of 128 or 256 lanes):
//   six 12-bit terms  slot | a << 8 | j << 10  in bits 0 .. 71,  then  pos | il << 7 (4 bits) | nterms << 11 | log2(group) << 14 | store << 16  in bits 72 .. 88
struct RowTablesS {
    const int* rec;      // [npos][rw]   GatherHdr | occupied slots (us / 4 words) | row offsets relative to the block (nbs + 1 words)
                         //              | first node-level CSR entry of every node's row (nbs words; the nodes of a block need not be
                         //              consecutive in memory: build_partition may form the blocks in a locality order)
    const unsigned* lanes;  // [npos][ls][3]   ls = the most lanes any position needs, rounded up to 32
    const int* vconn;    // [npos][vn + us]   the position's UNIQUE vertices (node ids; padded with its first one; vn = the most any position has,
                         //              rounded up to 32), then one word per slot: the four bytes index that list (k_build_row_verts_tet4).
                         //              A block of nine nodes sees ~180 elements but only ~70 distinct vertices: gathering every (slot, local
                         //              node) coordinate again at every position was 720 scattered 24-byte fetches and twelve LDS writes per lane
    const int* elem;     // [npos][us]
    const double* slotpar;  // [npos][us][2] (mu, lambda) of the element in each slot (piecewise-constant material), or null
    int rw, us, nbs, npos, ls, vn;
    int prio;            // s_setprio levels (round 5): bits 0-1 the two wavefronts that carry phase B (the last lanes), bits 2-3 the other two
};
constexpr int ROWS_TET4_VMAX = 256;   // unique vertices per position the tables can express (one per lane)
// the packed record <-> six 16-bit terms in three words + the control word
__host__ __device__ inline void rows_tet4_pack(const unsigned (&w3)[3], unsigned w, unsigned (&out)[3]) {
    const unsigned t0 = w3[0] & 0xfffu, t1 = (w3[0] >> 16) & 0xfffu, t2 = w3[1] & 0xfffu, t3 = (w3[1] >> 16) & 0xfffu, t4 = w3[2] & 0xfffu,
                   t5 = (w3[2] >> 16) & 0xfffu;
    out[0] = t0 | t1 << 12 | (t2 & 0xffu) << 24;
    out[1] = (t2 >> 8) | t3 << 4 | t4 << 16 | (t5 & 0xfu) << 28;
    out[2] = (t5 >> 4) | w << 8;
}

// ELEMPAR: (mu, lambda) per element: every term carries its element's pair, the block is
//   sum_e mu_e (tr G_e I + G_e^T) + lambda_e G_e  =  tr(Gmu) I + Gmu^T + Gla   with Gmu = sum mu_e G_e, Gla = sum lambda_e G_e
template <int OP, bool ELEMPAR = false, bool DBG = false>
__global__ void __launch_bounds__(256, 3) k_gather_rows_tet4(const KArgs a, const RowTablesS T) {
    constexpr int EK = FH_TET4, QC = 1, TL = 6;
    using E = ElemT<EK>;
    using O = OpT<OP, E::D>;
    constexpr int D = E::D, N = E::N, NG = E::NG, S = O::S;
    static_assert(D == 3 && N == 4 && NG == 4, "row-owner kernel: Tet4");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout L = make_layout<EK, OP, WHAT_MATRIX>(a.nq, a.ub, 0, a.nb_max, true, 0, 1, QC, 0, 2);
    double* lds = reinterpret_cast<double*>(smem);
    int* lds_i = reinterpret_cast<int*>(smem + sizeof(double) * (size_t)L.n_doubles);
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    stage_tables<EK>(a, L, lds);
    { const int pr = (T.prio >> ((tid >> 7) ? 0 : 2)) & 3; if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); }

    struct Rec { int w, vid, sw; };   // record word, vertex id and slot word of this lane
    const int npos = T.npos, vs = T.vn + T.us;
    const int p_begin = (int)((long long)blockIdx.x * npos / G), p_end = (int)((long long)(blockIdx.x + 1) * npos / G);
    auto load_rec = [&](int p, Rec& r) {
        p = min(p, npos - 1);
        r.w = T.rec[(size_t)p * T.rw + min(tid, T.rw - 1)];
        r.vid = T.vconn[(size_t)p * vs + min(tid, T.vn - 1)];   // (lanes behind the list re-read its last entry and park nothing)
        r.sw = T.vconn[(size_t)p * vs + T.vn + min(tid, T.us - 1)];
    };
    // lanes beyond the table's stride re-read its last record and switch themselves off
    struct LaneRec { unsigned x, y, z; };   // the packed record (see RowTablesS)
    auto load_lane = [&](int p) {
        const unsigned* q = T.lanes + ((size_t)min(p, npos - 1) * T.ls + min(tid, T.ls - 1)) * 3;
        LaneRec r{q[0], q[1], q[2]};
        if (tid >= T.ls) r.z = 0u;
        return r;
    };
    double V[D];
    auto load_verts = [&](const Rec& r) {
#pragma unroll
        for (int c = 0; c < D; ++c) V[c] = a.verts[(size_t)((a.ablate & 8) ? tid : r.vid) * D + c];   // (ablate bit 3: coalesced stand-in)
    };
    auto rec_base = [&](int parity) { return lds_i + parity * T.rw; };
    int* const slot_words = lds_i + 2 * T.rw;            // [us]: read by phase B only, like the vertex table
    auto park = [&](const Rec& r, int parity) {
        f64x2 xy;
        xy.x = V[0]; xy.y = V[1];
        double* row = lds + L.o_X + 4 * tid;             // vertex table: 32 bytes per entry, [x y | z -]
        if (tid < T.vn) {
            *reinterpret_cast<f64x2*>(row) = xy;
            row[2] = V[2];
        }
        if (tid < T.us) slot_words[tid] = r.sw;
        if (tid < T.rw) rec_base(parity)[tid] = r.w;
    };
    // (Round 6, measured and removed -- profiles/r06_c3_tables.txt: the finished blocks through an image of the position's rows in LDS and from
    // there to global memory by all 256 lanes, 16 contiguous bytes each, beside the next position's phase B: 0.61 against 0.47 ms.  The image's
    // own LDS traffic costs more than the straight stores do.)
    const double sqw = sqrt(a.qw[0]);
    int p = p_begin;
    if (p >= p_end) return;
    Rec nxt;
    LaneRec lane_cur;
    {
        Rec cur;
        load_rec(p, cur);
        load_verts(cur);
        lane_cur = load_lane(p);
        load_rec(p + 1, nxt);
        park(cur, 0);
        asm volatile("" : "+v"(nxt.w), "+v"(nxt.vid), "+v"(nxt.sw), "+v"(lane_cur.x), "+v"(lane_cur.z));
    }
    __syncthreads();

    // FENRIS_HIP_TRACE (instrumented instantiation): cycles per phase and wavefront, reported by fh_destroy under the pipelined kernel's names
    unsigned long long tr[6] = {0, 0, 0, 0, 0, 0}, tr_t = 0;
    auto stamp = [&](int k) {
        if constexpr (DBG) {
            const unsigned long long t = __builtin_readcyclecounter();
            tr[k] += t - tr_t;
            tr_t = t;
        }
    };
    if constexpr (DBG) tr_t = __builtin_readcyclecounter();
    int parity = 0;
    for (; p < p_end; ++p, parity ^= 1) {
        const bool have_next = (p + 1) < p_end;
        const int* rec = rec_base(parity);
        const GatherHdr hc = *reinterpret_cast<const GatherHdr*>(rec);
        const unsigned char* slot_b = reinterpret_cast<const unsigned char*>(rec + 8);
        const int* noff_l = rec + 8 + T.us / 4;
        load_verts(nxt);
        Rec nn;
        load_rec(p + 2, nn);
        LaneRec lane_nxt = load_lane(p + 1);
        const int U = (p == p_begin) ? hc.U : hc.k0;
        stamp(0);
        // phase B: one lane per new slot (one quadrature point)
        // ... on the LAST lanes of the workgroup: the row lanes fill the first wavefronts (162 of 256 lanes on a BCC mesh, the four-lane
        // groups of the diagonal blocks in wavefront 0), so the two phases load different SIMDs and the two workgroups of a CU overlap better
        const int bi = 255 - tid;
        if (bi < U && !(a.ablate & 1)) {   // (FENRIS_HIP_ABLATE bit 0, profiling: no phase B)
            const int u = (int)slot_b[bi];
            prologue<EK, OP, WHAT_MATRIX, true, true, false, true>(a, L, lds, lds_i, u, 0, T.elem + (size_t)p * T.us + u, 0, sqw,
                                                                   (unsigned)slot_words[u]);
        }
        stamp(1);
        lds_barrier();
        stamp(2);

        // phase C: G = sum over the lane's terms of h_a h_j^T
        const unsigned wl = lane_cur.z >> 8;
        const int nterms = (int)((wl >> 11) & 7u), grp = (int)((wl >> 14) & 3u);
        const unsigned tm[TL] = {lane_cur.x & 0xfffu, (lane_cur.x >> 12) & 0xfffu, (lane_cur.x >> 24) | (lane_cur.y & 0xfu) << 8,
                                 (lane_cur.y >> 4) & 0xfffu, (lane_cur.y >> 16) & 0xfffu, (lane_cur.y >> 28) | (lane_cur.z & 0xffu) << 4};
        double Gm[D][D], Gl[ELEMPAR ? D : 1][ELEMPAR ? D : 1];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                Gm[i][j] = 0.0;
                if (ELEMPAR) Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)] = 0.0;
            }
        // a wavefront without a single term (the lanes behind the last block of the position) skips the phase altogether
        if (__builtin_amdgcn_ballot_w64(nterms > 0) != 0ull) {
            unsigned pa[TL], pj[TL];
            double mu_t[ELEMPAR ? TL : 1], la_t[ELEMPAR ? TL : 1];
    #pragma unroll
            for (int t = 0; t < TL; ++t) {
                const unsigned term = tm[t];
                const double* pq = lds + L.o_QP + (size_t)(term & 255u) * L.qss;
                pa[t] = (unsigned)(unsigned long long)(pq + 4 * ((term >> 8) & 3u));
                pj[t] = (unsigned)(unsigned long long)(pq + 4 * ((term >> 10) & 3u));
                if constexpr (ELEMPAR) {  // fetched now (global memory), used after phase B
                    const double* sp = T.slotpar + 2 * ((size_t)p * T.us + (term & 255u));
                    mu_t[t] = sp[0];
                    la_t[t] = sp[1];
                }
            }
            // [x y] by ds_read_b128, z by ds_read_b64 (see k_gather_rows); two terms in flight while one is multiplied
            constexpr int AHEAD = 2, NB = AHEAD + 1;
            f64x2 av[NB], bv[NB];
            double az[NB], bz[NB];
            auto fetcht = [&](auto tk) {
                constexpr int tt = decltype(tk)::value, sl = tt % NB;
                av[sl] = lds_read_f64x2<0>(pa[tt]);
                az[sl] = lds_read_f64_at<16>(pa[tt]);
                bv[sl] = lds_read_f64x2<0>(pj[tt]);
                bz[sl] = lds_read_f64_at<16>(pj[tt]);
            };
            fetcht(std::integral_constant<int, 0>{});
            fetcht(std::integral_constant<int, 1>{});
            pipeline_consume<TL, D>([&](auto tk) {
                constexpr int tt = decltype(tk)::value, sl = tt % NB;
                constexpr int ahead = (TL - 1 - tt) < (AHEAD - 1) ? (TL - 1 - tt) : (AHEAD - 1);
                lds_wait<ahead * 4>();
                asm volatile("" : "+v"(av[sl]), "+v"(az[sl]), "+v"(bv[sl]), "+v"(bz[sl]));
                if constexpr (tt + AHEAD < TL) fetcht(std::integral_constant<int, tt + AHEAD>{});
                __builtin_amdgcn_sched_barrier(0);
                if (tt < nterms && !(a.ablate & 2)) {  // unused terms read slot 0: never let their values in  (ablate bit 1: no products)
                    const double ai[D] = {av[sl].x, av[sl].y, az[sl]};
                    const double bj[D] = {bv[sl].x, bv[sl].y, bz[sl]};
    #pragma unroll
                    for (int i = 0; i < D; ++i) {
                        const double am = ELEMPAR ? mu_t[ELEMPAR ? tt : 0] * ai[i] : ai[i];
                        const double al = ELEMPAR ? la_t[ELEMPAR ? tt : 0] * ai[i] : 0.0;
    #pragma unroll
                        for (int j = 0; j < D; ++j) {
                            Gm[i][j] = fma(am, bj[j], Gm[i][j]);
                            if (ELEMPAR) Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)] = fma(al, bj[j], Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)]);
                        }
                    }
                }
            });
            // Partial sums of a block meet by DPP exchanges in groups of 2 / 4 / 8 lanes.  Whether a lane takes part is a select on the
            // exchanged value, not a branch (nine guarded additions per stage cost an exec-mask pair each) and not a factor 0 / 1 either
            // (0 x NaN would carry a neighbour's NaN -- a non-finite vertex -- into blocks it does not belong to).  Every stage is skipped
            // by a wavefront none of whose lanes needs it (the third: a node with more than 24 elements).
            auto sel = [](bool take, double v) { return take ? v : 0.0; };
            if (__builtin_amdgcn_ballot_w64(grp >= 1) != 0ull) {   // every stage is skipped by a wavefront none of whose lanes needs it
    #pragma unroll
                for (int i = 0; i < D; ++i)
    #pragma unroll
                    for (int j = 0; j < D; ++j) {
                        Gm[i][j] += sel(grp >= 1, dpp_quad<0xB1>(Gm[i][j]));
                        if constexpr (ELEMPAR) Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)] =
                            Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)] + sel(grp >= 1, dpp_quad<0xB1>(Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)]));
                    }
            }
            if (__builtin_amdgcn_ballot_w64(grp >= 2) != 0ull) {
    #pragma unroll
                for (int i = 0; i < D; ++i)
    #pragma unroll
                    for (int j = 0; j < D; ++j) {
                        Gm[i][j] += sel(grp >= 2, dpp_quad<0x4E>(Gm[i][j]));
                        if constexpr (ELEMPAR) Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)] =
                            Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)] + sel(grp >= 2, dpp_quad<0x4E>(Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)]));
                    }
            }
            if (__builtin_amdgcn_ballot_w64(grp >= 3) != 0ull) {   // uniform over the wavefront
    #pragma unroll
                for (int i = 0; i < D; ++i)
    #pragma unroll
                    for (int j = 0; j < D; ++j) {
                        Gm[i][j] += sel(grp >= 3, dpp_xor4(Gm[i][j]));
                        if constexpr (ELEMPAR) Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)] =
                            Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)] + sel(grp >= 3, dpp_xor4(Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)]));
                    }
            }
        }
        stamp(3);
        // The requests of this position (vertices of the next block, records and lane words of the one after) are consumed HERE,
        // in front of the stores: loads and stores share one in-order counter, and behind the (branch-guarded) stores the wait
        // for these loads becomes vmcnt(0) -- every position then waited for its own nine stores per lane to drain, a write
        // latency per position (round 3: C3 0.63 -> see DESIGN 3.4).  X is read by phase B only, which lies behind the barrier above.
        if (have_next) park(nxt, parity ^ 1);
        asm volatile("" : "+v"(nn.w), "+v"(nn.vid), "+v"(nn.sw), "+v"(lane_nxt.x), "+v"(lane_nxt.z));
        if (((wl >> 16) & 1u) && !(a.ablate & 4)) {   // (ablate bit 2: no global stores)
            const int il = (int)((wl >> 7) & 15u), pos = (int)(wl & 127u);
            const int cnt = noff_l[il + 1] - noff_l[il];
            double* base = a.vals + (size_t)S * S * (size_t)(unsigned)noff_l[T.nbs + 1 + il] + S * pos;
            const bool ow = a.overwrite;
            double tr = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) tr += Gm[i][i];
            if (OP == FH_LAPLACE) {
                if (ow) base[0] = tr; else base[0] += tr;
            } else {
                // a row of the block is 24 bytes at an 8-byte boundary: one 16-byte and one 8-byte store
                typedef double f64x2_u8 __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double* row = base + (size_t)(i % S) * S * cnt;
                    double v[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        if constexpr (ELEMPAR) v[j] = ((i == j) ? tr : 0.0) + Gm[j][i] + Gl[i % (ELEMPAR ? D : 1)][j % (ELEMPAR ? D : 1)];
                        else v[j] = (i == j) ? fma(a.mu, tr + Gm[i][i], a.lambda * Gm[i][i]) : fma(a.mu, Gm[j][i], a.lambda * Gm[i][j]);
                    }
                    if (ow) {
                        f64x2_u8 lo;
                        lo.x = v[0]; lo.y = v[1];
                        *reinterpret_cast<f64x2_u8*>(row) = lo;
                        row[2 % S] = v[2];
                    } else {
#pragma unroll
                        for (int j = 0; j < D; ++j) row[j % S] += v[j];
                    }
                }
            }
        }
        nxt = nn;
        lane_cur = lane_nxt;
        stamp(4);
        lds_barrier();
        stamp(5);
    }
    if constexpr (DBG) {
        if (a.trace && (tid & 63) == 0) {
            unsigned long long* r = a.trace + 7 * (tid >> 6);
            for (int k = 0; k < 6; ++k) atomicAdd(r + k, tr[k]);
            atomicAdd(r + 6, 1ull);
        }
    }
}

// lanes of every position for the Tet4 kernel (see k_build_row_lanes).  Terms are kept in one compact array (a block has
// at most ms N of them) behind per-column offsets; blocks with up to 48 terms (a node of an unstructured mesh easily has
// 30-40 elements) take groups of up to 8 lanes.
// count_only: nothing is written but status[1] = the most lanes any position needs (atomicMax) -- the caller sizes the stride ls by it
static __global__ void __launch_bounds__(64) k_build_row_lanes_tet4(const int* p_rec, int rw_old, int us, int ms, int nbs, int npos, int rw_new,
                                                             int* rec_new, unsigned* lanes, int ls, int* status, const unsigned* row_real,
                                                             int count_only) {
    constexpr int N = 4, NKEY = 16 * 128, TMAX = 48, TL = 6, MAXTERMS = 1408;   // up to 16 nodes and 352 (node, element) entries per block
    __shared__ int cnt[NKEY], off[NKEY], fill[NKEY];
    __shared__ unsigned short terms[MAXTERMS];
    __shared__ unsigned lw[256][4];
    __shared__ int wave_tot[64];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int* rec = p_rec + (size_t)p * rw_old;
    const GatherHdr h = *reinterpret_cast<const GatherHdr*>(rec);
    const unsigned* ent = reinterpret_cast<const unsigned*>(rec + 8 + us / 4);
    const unsigned char* posb = reinterpret_cast<const unsigned char*>(rec + 8 + us / 4 + ms);
    const int* noff_old = rec + 8 + us / 4 + ms + ms * N / 4;
    int* out = rec_new + (size_t)p * rw_new;
    if (!count_only) {
        for (int i = lane; i < 8 + us / 4; i += 64) out[i] = rec[i];
        for (int i = lane; i <= nbs; i += 64) out[8 + us / 4 + i] = noff_old[i];
        for (int i = lane; i < nbs; i += 64) out[8 + us / 4 + nbs + 1 + i] = (i < h.nb) ? (int)row_real[h.i0 + i] : 0;
    }
    for (int i = lane; i < NKEY; i += 64) { cnt[i] = 0; fill[i] = 0; }
    for (int i = lane; i < 256 * 4; i += 64) lw[i / 4][i % 4] = 0u;
    __syncthreads();
    bool bad = h.m * N > MAXTERMS;
    for (int idx = lane; idx < h.m * N && !bad; idx += 64) {
        const int t = idx / N, j = idx % N;
        const unsigned e = ent[t];
        const unsigned il = e & 0xffu, pos = posb[t * N + j];
        if (il >= 16u || pos >= 128u || (e >> 16) >= 256u || ((e >> 8) & 0xffu) >= 4u) { bad = true; break; }
        atomicAdd(&cnt[(int)(il * 128u + pos)], 1);
    }
    __syncthreads();
    if (count_only) {   // lanes by classes: 25..48 terms -> 8 lanes, 13..24 -> 4, 7..12 -> 2, 1..6 -> 1 (as below)
        int need = 0;
        for (int base = 0; base < NKEY; base += 64) {
            const int Tn = cnt[base + lane];
            need += 8 * __popcll(__ballot(Tn > 4 * TL)) + 4 * __popcll(__ballot(Tn > 2 * TL && Tn <= 4 * TL)) +
                    2 * __popcll(__ballot(Tn > TL && Tn <= 2 * TL)) + __popcll(__ballot(Tn >= 1 && Tn <= TL));
        }
        if (lane == 0) atomicMax(status + 1, need);
        return;
    }
    // exclusive prefix sum of the counts: 16 consecutive keys per lane, then across the lanes
    {
        int local = 0;
        for (int k = 0; k < NKEY / 64; ++k) local += cnt[lane * (NKEY / 64) + k];
        wave_tot[lane] = local;
        __syncthreads();
        int before = 0;
        for (int l = 0; l < lane; ++l) before += wave_tot[l];
        for (int k = 0; k < NKEY / 64; ++k) {
            off[lane * (NKEY / 64) + k] = before;
            before += cnt[lane * (NKEY / 64) + k];
        }
    }
    __syncthreads();
    for (int idx = lane; idx < h.m * N && !bad; idx += 64) {
        const int t = idx / N, j = idx % N;
        const unsigned e = ent[t];
        const unsigned slot = e >> 16, a_loc = (e >> 8) & 0xffu, il = e & 0xffu, pos = posb[t * N + j];
        const int key = (int)(il * 128u + pos);
        const int s_ = atomicAdd(&fill[key], 1);
        terms[off[key] + s_] = (unsigned short)(slot | (a_loc << 8) | ((unsigned)j << 10));
    }
    __syncthreads();
    for (int key = lane; key < NKEY; key += 64) {  // fixed term order
        const int Tn = cnt[key];
        if (Tn > TMAX) bad = true;
        unsigned short* b = terms + off[key];
        for (int i = 1; i < Tn && !bad; ++i) {
            const unsigned short v = b[i];
            int k = i - 1;
            while (k >= 0 && b[k] > v) { b[k + 1] = b[k]; --k; }
            b[k + 1] = v;
        }
    }
    __syncthreads();
    // classes by lanes needed: 25..48 terms -> 8 lanes, 13..24 -> 4, 7..12 -> 2, 1..6 -> 1
    int n8 = 0, n4 = 0, n2 = 0, n1 = 0;
    for (int base = 0; base < NKEY; base += 64) {
        const int Tn = cnt[base + lane];
        n8 += __popcll(__ballot(Tn > 4 * TL));
        n4 += __popcll(__ballot(Tn > 2 * TL && Tn <= 4 * TL));
        n2 += __popcll(__ballot(Tn > TL && Tn <= 2 * TL));
        n1 += __popcll(__ballot(Tn >= 1 && Tn <= TL));
    }
    const int base4 = 8 * n8, base2 = base4 + 4 * n4, base1 = base2 + 2 * n2;
    const bool too_many = base1 + n1 > ls;  // status bit 1: the lane stride is too small (the host retries with 256)
    if (base1 + n1 > 256) bad = true;
    if (__ballot(bad || too_many)) {
        if (lane == 0) atomicOr(status, __ballot(bad) ? 1 : 2);
        for (int i = lane; i < 3 * ls; i += 64) lanes[(size_t)p * ls * 3 + i] = 0u;
        return;
    }
    int r8 = 0, r4 = 0, r2 = 0, r1 = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = 0; base < NKEY; base += 64) {
        const int key = base + lane;
        const int Tn = cnt[key];
        const unsigned il = (unsigned)key >> 7, pos = (unsigned)key & 127u;
        const unsigned short* b = terms + off[key];
        const unsigned long long m8 = __ballot(Tn > 4 * TL), m4 = __ballot(Tn > 2 * TL && Tn <= 4 * TL),
                                 m2 = __ballot(Tn > TL && Tn <= 2 * TL), m1 = __ballot(Tn >= 1 && Tn <= TL);
        auto lane_words = [&](int first, int Lidx, unsigned grp, bool leader) {
            const int n = max(0, min(TL, Tn - first));
            unsigned w3[3] = {0u, 0u, 0u};
            for (int t = 0; t < n; ++t) w3[t / 2] |= (unsigned)b[first + t] << (16 * (t % 2));
            lw[Lidx][0] = w3[0]; lw[Lidx][1] = w3[1]; lw[Lidx][2] = w3[2];
            lw[Lidx][3] = pos | (il << 7) | ((unsigned)n << 11) | (grp << 14) | (leader ? (1u << 16) : 0u);
        };
        if (Tn > 4 * TL) {
            const int L0 = 8 * (r8 + __popcll(m8 & below));
            for (int g = 0; g < 8; ++g) lane_words(TL * g, L0 + g, 3u, g == 0);
        } else if (Tn > 2 * TL) {
            const int L0 = base4 + 4 * (r4 + __popcll(m4 & below));
            for (int g = 0; g < 4; ++g) lane_words(TL * g, L0 + g, 2u, g == 0);
        } else if (Tn > TL) {
            const int L0 = base2 + 2 * (r2 + __popcll(m2 & below));
            for (int g = 0; g < 2; ++g) lane_words(TL * g, L0 + g, 1u, g == 0);
        } else if (Tn >= 1) {
            lane_words(0, base1 + r1 + __popcll(m1 & below), 0u, true);
        }
        r8 += __popcll(m8); r4 += __popcll(m4); r2 += __popcll(m2); r1 += __popcll(m1);
    }
    __syncthreads();
    for (int i = lane; i < ls; i += 64) {
        const unsigned w3[3] = {lw[i][0], lw[i][1], lw[i][2]};
        unsigned pk[3];
        rows_tet4_pack(w3, lw[i][3], pk);
        unsigned* q = lanes + ((size_t)p * ls + i) * 3;
        q[0] = pk[0]; q[1] = pk[1]; q[2] = pk[2];
    }
}

// With an element mask a block (I, J) of the pattern may have no active element and therefore no lane: k_gather_rows_tet4 writes every
// OWNED block exactly once and nothing else, so in overwrite mode the values of the node range are cleared first (masked contexts only:
// the multi-GPU partitions of fenris_amd/distributed.py are Hex8; found by tests/test_gpu_parity.py::test_overwrite_with_a_mask_...)
static __global__ void __launch_bounds__(256) k_zero_node_rows(const unsigned* noff, int n_lo, int n_hi, int ss, double* vals) {
    const size_t lo = (size_t)ss * noff[n_lo], hi = (size_t)ss * noff[n_hi];
    for (size_t k = lo + (size_t)blockIdx.x * 256 + threadIdx.x; k < hi; k += (size_t)gridDim.x * 256) vals[k] = 0.0;
}

// Unique vertices of every position for the Tet4 kernel (RowTablesS::vconn) from the per-slot connectivity of the pipelined tables
// (p_conn: [npos][us * 4], node 0 in empty slots).  One wavefront per position: the node ids go through a hash table in LDS, the occupied
// cells are numbered in table order, every slot gets the four numbers of its nodes.  Which cell a node lands in may depend on the order
// the lanes arrive (linear probing): only the numbering inside the position's vertex table does, never a value of the matrix.
// More than ROWS_TET4_VMAX distinct vertices: status bit 2 (the caller keeps the pipelined kernel).
// vn: length of a position's vertex list in the table (a multiple of 32, <= ROWS_TET4_VMAX).  count_only: nothing is written but status[1] = the most
// distinct vertices any position has (atomicMax).
static __global__ void __launch_bounds__(64) k_build_row_verts_tet4(const int* p_conn, int us, int npos, int* vconn, int* status, int vn, int count_only) {
    constexpr int H = 2048, CSMAX = 1024;
    __shared__ int key[H];
    __shared__ unsigned short cell_of[CSMAX], num[H];
    const int p = blockIdx.x, lane = threadIdx.x, cs = 4 * us, vs = vn + us;
    const int* conn = p_conn + (size_t)p * cs;
    int* out = vconn + (size_t)p * vs;
    for (int i = lane; i < H; i += 64) key[i] = -1;
    __syncthreads();
    for (int i = lane; i < cs; i += 64) {
        const int v = conn[i];
        unsigned h = ((unsigned)v * 2654435761u) >> 21;   // 11 bits
        for (;;) {
            const int old = atomicCAS(&key[h], -1, v);
            if (old == -1 || old == v) break;
            h = (h + 1) & (H - 1);
        }
        cell_of[i] = (unsigned short)h;
    }
    __syncthreads();
    int total = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = 0; base < H; base += 64) {
        const bool occ = key[base + lane] != -1;
        const unsigned long long m = __ballot(occ);
        const int idx = total + __popcll(m & below);
        num[base + lane] = (unsigned short)idx;
        if (!count_only && occ && idx < vn) out[idx] = key[base + lane];
        total += __popcll(m);
    }
    if (count_only) {
        if (lane == 0) atomicMax(status + 1, total);
        return;
    }
    if (total > vn) {
        if (lane == 0) atomicOr(status, 4);
        return;
    }
    __syncthreads();
    const int first = (cs > 0) ? conn[0] : 0;
    for (int i = total + lane; i < vn; i += 64) out[i] = first;   // padding: a vertex this position fetches anyway
    for (int s_ = lane; s_ < us; s_ += 64) {
        unsigned w = 0;
        for (int g = 0; g < 4; ++g) w |= (unsigned)num[cell_of[4 * s_ + g]] << (8 * g);
        out[vn + s_] = (int)w;
    }
}

}  // namespace fenris_hip
