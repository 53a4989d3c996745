// Affine-element owner-computes stiffness kernel, fourth form (k_affine_rows_dma): k_affine_rows (affine_rows.hip) with a loader wave
// that moves the element records by LDS-DMA (global_load_lds) FOUR positions ahead.
//
// Why (round 3, scripts/ab_in_context.py, one context, same buffers; ms with 0.26 of k_affine_records): the kernel 4.62; without any
// record fetch 4.16; records from a cache-resident table AND no element-id fetch 4.25, either of the two alone 4.56 / 4.64.  The loader's
// chain  element ids -> record addresses -> records  runs two positions ahead in registers (three spill: the register budget is the
// row lanes'), and under the store stream a fetch takes 2.3 us and more: the loader, not the memory system, paces the position.
// Here the records go from global memory straight into their LDS stage (no registers: the depth is a matter of LDS, 2.5 KB per
// stage), the element ids likewise into a small ring eight positions ahead, the headers come by scalar loads, and the lane record of
// a new table is fetched by the row lanes themselves one position ahead -- the loader issues exactly four DMA operations per position
// and nothing else on the vector-memory counter, so that one counted  s_waitcnt vmcnt  per position says "the records of the next
// position have landed" (DMA completes in order).
//
//   queue (oldest first) at the top of step i:   ... rec(i+1) ids(i+1+D) rec(i+2) ids(i+2+D) ... rec(i+D-1) ids(i+2D-1)
//   step i:  wait vmcnt(1 + (D-2)(R+1))  [rec(i+1) landed, hence ids(i+D)];  ds_read ids(i+D);  issue rec(i+D), ids(i+2D);  barrier
//   stages:  position q uses record stage q mod (D+1) and id slot q mod 16.
#include <hip/hip_runtime.h>

#include "affine_rows.hpp"
#include "small_ops.hpp"

namespace fenris_hip {

namespace {
constexpr int DMA_D = 4, NS = DMA_D + 1, NI = 16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int dma_i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) dma_i32x4* const_hdr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS-DMA, hand-issued (cdna_hip_programming.md 5.7): lane l of the active lanes copies 16 (4) bytes from its global address to
// LDS byte address lds_dst + 16 l (4 l); lds_dst is wavefront-uniform and goes through M0, which is saved and restored.  hipcc does
// not count these operations (the builtin it does count, and it then drains ALL of them -- vmcnt(0) -- in front of every LDS access
// and every asm statement of the wave): their completion is counted by hand, see the loader.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
}  // namespace

size_t affine_dma_lds_bytes(int op, int us, int acc_max) {
    const int gw = (op == FH_LAPLACE) ? AFFINE_ROWS_GW_LAP : AFFINE_ROWS_GW_LE;
    const size_t accp = (size_t)((acc_max + 16 + 1) & ~1);
    return sizeof(double) * ((size_t)65 * gw + (size_t)NS * us * gw + 2 * accp) + 4 * sizeof(int4) + NI * 32 * sizeof(int);
}

template <int OP, bool OVERWRITE>
__global__ void __launch_bounds__(384, 5) k_affine_rows_dma(const KArgs a, const AffineRowTables T, const int ablate_arg) {
    constexpr bool LAP = (OP == FH_LAPLACE);
    constexpr int S = LAP ? 1 : 3, SS = S * S;
    constexpr int GW = LAP ? AFFINE_ROWS_GW_LAP : AFFINE_ROWS_GW_LE;
    const bool nt_stores = (ablate_arg & AFFINE_ROWS_NT_STORES) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* GH = reinterpret_cast<double*>(smem);   // [65][GW]
    double* JS = GH + 65 * GW;                      // [NS][us][GW]  record stages, written by DMA
    const int accp = (T.acc_max + 16 + 1) & ~1;
    double* OUT = JS + NS * T.us * GW;              // [2][accp]
    int4* HDR = reinterpret_cast<int4*>(OUT + 2 * accp);  // [4] ring of position headers {first value, rows, flags | changed << 2 | id << 8, head}
    int* IDS = reinterpret_cast<int*>(HDR + 4);     // [NI][32] element ids of the slots, written by DMA

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, npos = T.npos_all;
    const int p_begin = T.pos0 + (int)((long long)blockIdx.x * T.npos / G), p_end = T.pos0 + (int)((long long)(blockIdx.x + 1) * T.npos / G);
    if (p_begin >= p_end) return;
    for (int i = tid; i < 65 * GW; i += 384) GH[i] = (i < 64 * GW) ? T.ghat[i] : 0.0;
    for (int i = tid; i < 2 * accp; i += 384) OUT[i] = 0.0;
    const size_t vals_w = reinterpret_cast<size_t>(a.vals) >> 3;
    auto head_of = [&](int r0) { return (int)((vals_w + (size_t)SS * (size_t)r0) & 15); };

    if (wave >= 5) {
        // ------------------------------------------------------------------------------------------ store wave(s)
        // NSTORE wavefronts share the work as one unit of SL = 64 NSTORE lanes: a trip moves SL consecutive 16-byte pieces.
        constexpr int SL = 64;
        const int lane = tid - 320;
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
            // non-temporal stores (FENRIS_HIP_AFFINE_NT; default: Laplace only): the rows are written once and never read by this
            // kernel.  3 % on Laplace; on elasticity equal within the run-to-run spread.
            if constexpr (OVERWRITE) { if (nt_stores) __builtin_nontemporal_store(val, dst); else *dst = val; }
            else { const f64x2 o = *dst; f64x2 r; r.x = o.x + val.x; r.y = o.y + val.y; *dst = r; }
        };
        auto put1 = [&](double* dst, double val) {
            if constexpr (OVERWRITE) { if (nt_stores) __builtin_nontemporal_store(val, dst); else *dst = val; } else *dst += val;
        };
        auto stream_out = [&](const int4 hv, double* buf, double* other, bool carry_in, bool carry_out) {
            const int r0 = rfl(hv.x), nrow = rfl(hv.y), flags = rfl(hv.z), head = rfl(hv.w) & 15;
            double* line0 = a.vals + (size_t)SS * (size_t)r0 - head;
            const int lo = carry_in ? 0 : head, hi = head + SS * nrow;
            const int L = carry_out ? (hi & ~15) : hi;          // stored now: [lo, L); carried: [L, hi)
            const bool zero = !(flags & 1);
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
            if (zero) {  // some (node, column) block of these rows has no owner lane (element masks): clear what was read --
                         // every lane the pieces it fetched itself, the first lanes the ends and the carried part
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f64x2 z2 = {0.0, 0.0};
                for (int t = 0; t < nfull; ++t) b2[SL * t] = z2;
                if (lane < rem) b2[SL * nfull] = z2;
                if (lane == 0 && e_lo >= 0) buf[e_lo] = 0.0;
                if (lane == 0 && e_hi >= 0) buf[e_hi] = 0.0;
                if (lane < hi - L) buf[L + lane] = 0.0;
                if (lane < 2 * k0 - lo && lo + lane != e_lo) buf[lo + lane] = 0.0;   // nothing: [lo, 2 k0) is e_lo alone
            }
        };
        lds_barrier();  // B0
        bool carry_in = false;
        int par = 0;
        for (int p = p_begin; p < p_end; ++p, par ^= 1) {
            if (p > p_begin) {
                const int4 h_prev = HDR[(p - 1) & 3];
                const int r0_cur = rfl(HDR[p & 3].x);
                const bool carry_out = r0_cur == rfl(h_prev.x) + rfl(h_prev.y);
                stream_out(h_prev, OUT + (size_t)(par ^ 1) * accp, OUT + (size_t)par * accp, carry_in, carry_out);
                carry_in = carry_out;
            }
            lds_barrier();
        }
        stream_out(HDR[(p_end - 1) & 3], OUT + (size_t)(par ^ 1) * accp, OUT + (size_t)par * accp, carry_in, false);
        return;
    }


    if (wave == 4) {
        // ------------------------------------------------------------------------------------------ loader wave (DMA)
        const int lane = tid - 256;
        constexpr int NPC = GW / 2;                    // 16-byte pieces per record
        constexpr int ROUNDS = (NPC * 32 + 63) / 64;   // us <= 32 slots
        const int npieces = NPC * T.us;
        auto slot_of = [&](int r) { return min(lane + 64 * r, npieces - 1) / NPC; };
        auto piece_of = [&](int r) { const int i = min(lane + 64 * r, npieces - 1); return i - (i / NPC) * NPC; };
        const const_hdr_t hdrs = (const_hdr_t)T.hdr;   // wavefront-uniform, read-only: scalar loads
        auto ring_entry = [&](int q) {
            // (the index through readfirstlane: only a provably uniform index makes these scalar loads -- a vector load here would put
            // the compiler's own vmcnt(0) into every step and drain the DMA queue)
            const dma_i32x4 hq = hdrs[__builtin_amdgcn_readfirstlane(min(q, npos - 1))];
            const int id = hq.z >> 8;
            const bool changed = (q == p_begin) || id != (hdrs[__builtin_amdgcn_readfirstlane(min(max(q - 1, 0), npos - 1))].z >> 8);
            int4 o;
            o.x = hq.x; o.y = hq.y; o.z = (hq.z & 1) | (changed ? 4 : 0) | (id << 8); o.w = head_of(hq.x) | (hq.w << 8);
            return o;
        };
        const unsigned ids_addr = (unsigned)(unsigned long long)IDS, hdr_lds = (unsigned)(unsigned long long)HDR;
        // ids of position q -> IDS[q mod NI] (one DMA, lanes < us); records of position q -> JS[(q - p_begin) mod NS] (ROUNDS DMAs)
        auto dma_ids = [&](int q) {
            const int qc = min(q, npos - 1);
            if (lane < T.us)
                dma4(T.elem + (size_t)(unsigned)qc * (unsigned)T.us + lane, __builtin_amdgcn_readfirstlane(ids_addr + (unsigned)((q & (NI - 1)) * 32 * 4)));
        };
        auto lds_int = [&](unsigned addr) { int v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr)); return v; };
        auto dma_records = [&](int q) {
            const unsigned ids = ids_addr + (unsigned)((q & (NI - 1)) * 32 * 4);
            double* stage = JS + (size_t)((q - p_begin) % NS) * T.us * GW;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int e = max(lds_int(ids + 4u * (unsigned)slot_of(r)), 0);
                const f64x2* src = reinterpret_cast<const f64x2*>(T.rec) + (size_t)(unsigned)e * NPC + piece_of(r);
                if (lane + 64 * r < npieces)
                    dma16(src, __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(reinterpret_cast<f64x2*>(stage) + 64 * r)));
            }
        };
        {
            const int4 e0 = ring_entry(p_begin), e1 = ring_entry(p_begin + 1);
            if (lane == 0) { HDR[p_begin & 3] = e0; HDR[(p_begin + 1) & 3] = e1; }
        }
        // prologue in the order of the steady state: ids of the first D positions, then  rec(j) ids(j + D)  for j = 0 .. D - 1
#pragma unroll
        for (int j = 0; j < DMA_D; ++j) dma_ids(p_begin + j);
        wait_vmcnt<0>();
#pragma unroll
        for (int j = 0; j < DMA_D; ++j) { dma_records(p_begin + j); dma_ids(p_begin + j + DMA_D); }
        wait_vmcnt<1 + (DMA_D - 1) * (ROUNDS + 1)>();   // rec(0) landed
        lds_barrier();  // B0
        for (int p = p_begin; p < p_end; ++p) {
            wait_vmcnt<1 + (DMA_D - 2) * (ROUNDS + 1)>();   // rec(p + 1) landed, and with it the ids of p + D (older)
            dma_records(p + DMA_D);
            dma_ids(p + 2 * DMA_D);
            const int4 e2 = ring_entry(p + 2);
            if (lane == 0) {
                const dma_i32x4 ev = {e2.x, e2.y, e2.z, e2.w};
                asm volatile("ds_write_b128 %0, %1" ::"v"(hdr_lds + 16u * (unsigned)((p + 2) & 3)), "v"(ev));
            }
            lds_barrier();
        }
        wait_vmcnt<0>();   // nothing of this wave may land in LDS after the workgroup is gone
        return;
    }

    // ---------------------------------------------------------------------------------------------- row waves
    // Element records and headers come through LDS; the lane record of a new table is fetched from global memory one position ahead.
    lds_barrier();  // B0
    const unsigned hdr_addr = (unsigned)(unsigned long long)HDR + 8u;   // .z (flags | slot << 1 | changed << 2 | id << 8), .w (head)
    uint2 lane_cur = {0u, 0u}, lane_nxt = {0u, 0u};
    auto table_entry = [&](int z_) { return T.lanes[(size_t)(unsigned)(z_ >> 8) * 256u + (unsigned)tid]; };
    f64x2 gq0[3] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}}, gq1[3] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};   // Laplace: Ghat of the lane's terms
    int par = 0;
    for (int p = p_begin; p < p_end; ++p, par ^= 1) {
        int zw[2], zn[2];
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(*reinterpret_cast<long long*>(zw)), "=&v"(*reinterpret_cast<long long*>(zn))
                     : "v"(hdr_addr + 16u * (unsigned)(p & 3)), "v"(hdr_addr + 16u * (unsigned)((p + 1) & 3)) : "memory");
        const int z = __builtin_amdgcn_readfirstlane(zw[0]);
        const int z_next = __builtin_amdgcn_readfirstlane(zn[0]);
        const int head = zw[1] & 15;
        if (z & 4) {   // the lane table changed with this position: its record was requested a position ago (the first one: now)
            lane_cur = (p == p_begin) ? table_entry(z) : lane_nxt;
            if constexpr (LAP) {  // ... and with it the reference blocks of this lane's two terms: kept in registers
                const f64x2* q0 = reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(GH) + ((lane_cur.x >> 5) & 127u) * (GW * 8));
                const f64x2* q1 = reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(GH) + ((lane_cur.x >> 17) & 127u) * (GW * 8));
#pragma unroll
                for (int h = 0; h < 3; ++h) { gq0[h] = q0[h]; gq1[h] = q1[h]; }
            }
        }
        if ((z_next & 4) && p + 1 < p_end) lane_nxt = table_entry(z_next);   // lands while this position is computed

        const unsigned x = lane_cur.x, y = lane_cur.y;
        const char* js = reinterpret_cast<const char*>(JS + (size_t)((p - p_begin) % NS) * T.us * GW);
        const char* gh = reinterpret_cast<const char*>(GH);
        const unsigned oR0 = (x & 31u) * (GW * 8), oG0 = ((x >> 5) & 127u) * (GW * 8);
        const unsigned oR1 = ((x >> 12) & 31u) * (GW * 8), oG1 = ((x >> 17) & 127u) * (GW * 8);
        const int grp = (int)((x >> 24) & 3u);
        char* out_par = reinterpret_cast<char*>(OUT + (size_t)par * accp);
        if constexpr (LAP) {
            const f64x2* m0 = reinterpret_cast<const f64x2*>(js + oR0);
            const f64x2* m1 = reinterpret_cast<const f64x2*>(js + oR1);
            double s = 0.0;
            {
#pragma unroll
                for (int h = 0; h < 3; ++h) { const f64x2 m = m0[h], g = gq0[h]; s = fma(g.x, m.x, s); s = fma(g.y, m.y, s); }
#pragma unroll
                for (int h = 0; h < 3; ++h) { const f64x2 m = m1[h], g = gq1[h]; s = fma(g.x, m.x, s); s = fma(g.y, m.y, s); }
            }
            if (grp >= 1) s += dpp_quad_full<0xB1>(s);
            if (grp >= 2) s += dpp_quad_full<0x4E>(s);
            if ((x >> 28) & 1u) *reinterpret_cast<double*>(out_par + 8 * head + (y & 0xffffu)) = s;
        } else {
            double H[3][3];
            {
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
        lds_barrier();
    }
}


template <int OP>
static auto affine_dma_pick(bool ow) -> void (*)(const KArgs, const AffineRowTables, int) {
    return ow ? k_affine_rows_dma<OP, true> : k_affine_rows_dma<OP, false>;
}

hipError_t affine_dma_launch(int op, int grid, size_t lds_bytes, hipStream_t stream, const KArgs& a, const AffineRowTables& T, int flags) {
    const bool ow = a.overwrite != 0;
    void (*kern)(const KArgs, const AffineRowTables, int) = op == FH_LAPLACE ? affine_dma_pick<FH_LAPLACE>(ow) : affine_dma_pick<FH_LINEAR_ELASTIC>(ow);
    if (lds_bytes > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(384), lds_bytes, stream, a, T, flags);
    return hipGetLastError();
}

}  // namespace fenris_hip
