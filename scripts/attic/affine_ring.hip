// Affine-element owner-computes stiffness kernel, third form (k_affine_ring): the arithmetic, the tables and the roles of
// k_affine_rows (affine_rows.hip: four row waves, one loader wave, one store wave per workgroup, persistent over a contiguous
// range of positions in CSR order), but NO workgroup barrier inside the sweep.  The waves hand work to each other through
// monotonic counters in LDS, and the finished rows are staged in a RING that mirrors the value stream:
//
//  * ring: RING doubles (a power of two), ring offset rho of a value = its distance from the 128-byte line boundary below the
//    first value of the current run of consecutive positions, taken mod RING.  A position's rows continue where the previous
//    position's ended, so an incomplete last line simply stays in place until the next position completes it (the carry copy of
//    the second form is gone), and the row waves may run as far ahead of the store wave as the ring has room -- with one
//    barrier per position their idle times added up: the store wave sat at the barrier 22 % of its time while the memory
//    system was the limit the other 78 %.
//  * counters (FL[]): rows[w] = positions finished by row wave w; load = positions whose records, lane table and header (and
//    the header of the following position) are in LDS; cons = ring offset up to which the store wave has fetched the staged
//    rows; spos = positions the store wave has finished.
//      row wave, position i:   waits load >= i + 1 and end(i) - cons <= RING;  publishes rows[w] = i + 1
//      store wave, position i: waits min rows >= i + 1;                        publishes cons, spos = i + 1
//      loader wave, step i:    waits min rows >= i + 2 - DJ (the record stage it overwrites), spos >= i - 5 (the header entry
//                              it overwrites) and, before it parks a lane table, that the row waves have left the table that
//                              occupied the slot;  publishes load = i + 2
//    Every wait is on a strictly earlier position of another role: no cycle.  A wave publishes with its own LDS traffic
//    drained (s_waitcnt lgkmcnt(0)) and a reader issues its data fetches after the counter arrived, so no fence is needed
//    inside one CU's LDS.
//
// Exact symmetry and run-to-run reproducibility are those of k_affine_rows: same lane tables, same order of the terms.
#include <hip/hip_runtime.h>

#include "affine_rows.hpp"
#include "small_ops.hpp"

namespace fenris_hip {

namespace {
constexpr int DJ = 4;   // stages of slot records (position i uses stage i mod DJ)
constexpr int NH = 8;   // header ring
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ i32x4 lds_poll4(unsigned addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_put1(unsigned addr, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_put2(unsigned addr, i32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
}  // namespace

int affine_ring_doubles(int acc_max, int want_kb) {
    int ring = 1024;
    while (ring < 2 * (acc_max + 16)) ring *= 2;
    while (want_kb > 0 && ring * 8 < want_kb * 1024) ring *= 2;
    return ring;
}

size_t affine_ring_lds_bytes(int op, int us, int ring) {
    const int gw = (op == FH_LAPLACE) ? AFFINE_ROWS_GW_LAP : AFFINE_ROWS_GW_LE;
    const size_t head = (sizeof(double) * ((size_t)65 * gw + (size_t)DJ * us * gw) + 15) & ~(size_t)15;
    return head + sizeof(double) * (size_t)ring + NH * 16 + 2 * 256 * sizeof(uint2) + 32;
}

template <int OP, bool OVERWRITE, bool DBG, int DEPTH>
__global__ void __launch_bounds__(384, 5) k_affine_ring(const KArgs a, const AffineRowTables T, const int ring, const int ablate_arg) {
    constexpr bool LAP = (OP == FH_LAPLACE);
    constexpr int S = LAP ? 1 : 3, SS = S * S;
    constexpr int GW = LAP ? AFFINE_ROWS_GW_LAP : AFFINE_ROWS_GW_LE;
    const int ablate = DBG ? (ablate_arg & 0xffff) : 0;
    const bool nt_stores = (ablate_arg & AFFINE_ROWS_NT_STORES) != 0;
    const int throttle = (ablate_arg >> 20) & 0xff;   // store wave: at most this many stores in flight before the next group (0: no limit)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* GH = reinterpret_cast<double*>(smem);   // [65][GW]
    double* JS = GH + 65 * GW;                      // [DJ][us][GW]
    char* OUTc = smem + ((sizeof(double) * ((size_t)65 * GW + (size_t)DJ * T.us * GW) + 15) & ~(size_t)15);   // ring
    i32x4* HDR = reinterpret_cast<i32x4*>(OUTc + (size_t)ring * 8);  // [NH] {r0, nrow, flags | slot << 1 | changed << 2 | continues << 3, rho}
    uint2* LT = reinterpret_cast<uint2*>(HDR + NH);                   // [2][256]
    int* FL = reinterpret_cast<int*>(LT + 512);                       // rows[4], cons, spos, load, -
    const unsigned maskb = (unsigned)ring * 8u - 1u;

    const int tid = threadIdx.x;
    const int wave = rfl(tid >> 6);
    const int G = gridDim.x, npos = T.npos_all;
    const int p_begin = T.pos0 + (int)((long long)blockIdx.x * T.npos / G), p_end = T.pos0 + (int)((long long)(blockIdx.x + 1) * T.npos / G);
    const int n = p_end - p_begin;
    if (n <= 0) return;
    for (int i = tid; i < 65 * GW; i += 384) GH[i] = (i < 64 * GW) ? T.ghat[i] : 0.0;
    for (int i = tid; i < ring; i += 384) reinterpret_cast<double*>(OUTc)[i] = 0.0;
    if (tid < 8) FL[tid] = (tid == 6) ? 1 : 0;   // load = 1: the loader's prologue provides position 0 before the first barrier
    const size_t vals_w = reinterpret_cast<size_t>(a.vals) >> 3;
    auto head_of = [&](int r0) { return (unsigned)((vals_w + (size_t)SS * (size_t)r0) & 15); };
    const unsigned fl_addr = (unsigned)(unsigned long long)FL;
    const unsigned hdr_addr = (unsigned)(unsigned long long)HDR;
    // FENRIS_HIP_TRACE (instrumented instantiation): cycles per role and segment, summed over workgroups: trace[7 role + k], [7 role + 6] = waves
    unsigned long long tr[4] = {0, 0, 0, 0}, tr_t = 0;
    const bool tracing = DBG && a.trace != nullptr;
    auto tr_mark = [&](int k) { if (tracing) { const unsigned long long t = __builtin_readcyclecounter(); tr[k] += t - tr_t; tr_t = t; } };
    auto tr_report = [&](int role) {
        if (tracing && (tid & 63) == 0) {
            for (int k = 0; k < 4; ++k) atomicAdd(a.trace + 7 * role + k, tr[k]);
            atomicAdd(a.trace + 7 * role + 6, 1ull);
            if (role == 0) a.trace[30] = 0x52494E47ull;   // "RING": labels of the report
        }
    };

    if (wave == 5) {
        // ------------------------------------------------------------------------------------------ store wave
        const int lane = tid - 320;
        auto put = [&](f64x2* dst, f64x2 val) {
            if (DBG && (ablate & 1)) return;
            if constexpr (OVERWRITE) { if (nt_stores) __builtin_nontemporal_store(val, dst); else *dst = val; }
            else { const f64x2 o = *dst; f64x2 r; r.x = o.x + val.x; r.y = o.y + val.y; *dst = r; }
        };
        auto put1 = [&](double* dst, double val) {
            if (DBG && (ablate & 1)) return;
            if constexpr (OVERWRITE) { if (nt_stores) __builtin_nontemporal_store(val, dst); else *dst = val; } else *dst += val;
        };
        auto ring2 = [&](unsigned k) { return reinterpret_cast<f64x2*>(OUTc + ((k << 4) & maskb)); };          // 16-byte piece k
        auto ring1 = [&](unsigned v) { return reinterpret_cast<double*>(OUTc + ((v << 3) & maskb)); };         // value v
        lds_barrier();  // B0
        if (ablate_arg & (1 << 28)) __builtin_amdgcn_s_setprio(3);   // FENRIS_HIP_AFFINE_PRIO bit 0: the store wave issues first
        double* seg_g = nullptr;      // global address of ring offset seg_rho0 (a line boundary)
        unsigned seg_rho0 = 0, rho_done = 0;
        if (tracing) tr_t = __builtin_readcyclecounter();
        for (int i = 0; i < n; ++i) {
            for (;;) {
                const i32x4 r = lds_poll4(fl_addr);
                if (min(min(rfl(r.x), rfl(r.y)), min(rfl(r.z), rfl(r.w))) >= i + 1) break;
                __builtin_amdgcn_s_sleep(1);
            }
            tr_mark(1);
            const i32x4 hv = HDR[i & (NH - 1)];
            const int r0 = rfl(hv.x), nrow = rfl(hv.y), z = rfl(hv.z);
            const unsigned start = (unsigned)rfl(hv.w), end = start + (unsigned)(SS * nrow);
            const bool next_cont = (i + 1 < n) && ((rfl(HDR[(i + 1) & (NH - 1)].z) & 8) != 0);
            unsigned lo = rho_done;
            if (i == 0 || !(z & 8)) {   // a new run of consecutive positions: its first line may be partial
                const unsigned head = start & 15u;
                seg_g = a.vals + (size_t)SS * (size_t)r0 - head;
                seg_rho0 = start - head;
                lo = start;
            }
            unsigned L = next_cont ? (end & ~15u) : end;     // stored now: [lo, L); an incomplete last line waits for the next position
            if ((int)(L - lo) < 0) L = lo;
            const bool zero = T.incomplete != 0;   // (a later position may reuse the place of a complete one: all or nothing)
            const unsigned k0 = (lo + 1u) >> 1, k1 = L >> 1;    // whole 16-byte pieces [k0, k1)
            const int np = max((int)(k1 - k0), 0);
            const int nfull = np >> 6, rem = np & 63;
            f64x2* gout = reinterpret_cast<f64x2*>(seg_g) + (int)(k0 - (seg_rho0 >> 1)) + lane;
            const unsigned kl = k0 + (unsigned)lane;
            auto pace = [&]() {   // keep the queue of this CU's vector memory path short: the loader's fetches wait behind the stores
                switch (throttle) {
                    case 0: break;
                    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
                }
            };
            int t = 0;
            for (; t + 4 <= nfull; t += 4) {
                pace();
                const f64x2 v0 = *ring2(kl + 64u * t), v1 = *ring2(kl + 64u * (t + 1)), v2 = *ring2(kl + 64u * (t + 2)), v3 = *ring2(kl + 64u * (t + 3));
                put(gout + 64 * t, v0); put(gout + 64 * (t + 1), v1); put(gout + 64 * (t + 2), v2); put(gout + 64 * (t + 3), v3);
            }
            for (; t < nfull; ++t) put(gout + 64 * t, *ring2(kl + 64u * t));
            if (lane < rem) put(gout + 64 * nfull, *ring2(kl + 64u * nfull));
            const bool has_lo = (lo & 1u) && (int)(L - lo) > 0;               // the ends of a run of positions: single doubles
            const bool has_hi = (L & 1u) && (int)(L - 1u - lo) >= 0 && (int)(L - lo) > 0;
            if (has_lo && lane == 0) put1(seg_g + (lo - seg_rho0), *ring1(lo));
            if (has_hi && lane == 0 && !(has_lo && L - 1u == lo)) put1(seg_g + (L - 1u - seg_rho0), *ring1(L - 1u));
            if (zero) {  // some (node, column) block of these rows has no owner lane (element masks): clear what was fetched
                lds_drain();
                const f64x2 z2 = {0.0, 0.0};
                for (int u = 0; u < nfull; ++u) *ring2(kl + 64u * u) = z2;
                if (lane < rem) *ring2(kl + 64u * nfull) = z2;
                if (has_lo && lane == 0) *ring1(lo) = 0.0;
                if (has_hi && lane == 0) *ring1(L - 1u) = 0.0;
            }
            rho_done = L;
            const unsigned cons = next_cont ? L : ((end + 15u) & ~15u);
            tr_mark(0);
            lds_drain();
            if (lane == 0) { i32x2 w; w.x = (int)cons; w.y = i + 1; lds_put2(fl_addr + 16u, w); }
            tr_mark(2);
        }
        tr_report(2);
        return;
    }

    if (wave == 4) {
        // ------------------------------------------------------------------------------------------ loader wave
        // Every global load of the kernel, as in k_affine_rows: the element records of the next position's slots, the lane
        // table when it changes, the position headers; requests run DEPTH positions ahead of their use.
        const int lane = tid - 256;
        constexpr int NPC = GW / 2;                    // 16-byte pieces per record
        constexpr int ROUNDS = (NPC * 32 + 63) / 64;   // us <= 32 slots
        const int npieces = NPC * T.us;
        auto slot_of = [&](int r) { return min(lane + 64 * r, npieces - 1) / NPC; };
        auto piece_of = [&](int r) { const int i = min(lane + 64 * r, npieces - 1); return i - (i / NPC) * NPC; };
        // profiling only (instrumented instantiation): 128 every record from the first 4096 (cache-resident), 256 non-temporal record
        // fetches, 512 only the first round of record fetches, 1024 no element-id fetches
        auto load_elem = [&](int p, int r) {
            if (DBG && (ablate & 1024)) return (p * 32 + slot_of(r)) & 0xfffff;
            return T.elem[(size_t)((unsigned)min(p, npos - 1) * (unsigned)T.us + (unsigned)slot_of(r))];
        };
        auto load_piece = [&](int e, int r) {
            if (DBG && (ablate & 512) && r > 0) { f64x2 z = {0.0, 0.0}; return z; }
            const f64x2* q = reinterpret_cast<const f64x2*>(T.rec) + (size_t)(unsigned)((DBG && (ablate & 128)) ? (max(e, 0) & 4095) : max(e, 0)) * NPC + piece_of(r);
            if (DBG && (ablate & 256)) return __builtin_nontemporal_load(q);
            return *q;
        };
        auto park_piece = [&](int stage, int r, f64x2 v) {
            if (lane + 64 * r < npieces) reinterpret_cast<f64x2*>(JS + ((size_t)stage * T.us + slot_of(r)) * GW)[piece_of(r)] = v;
        };
        auto load_tab = [&](int id, int half) { return reinterpret_cast<const uint4*>(T.lanes)[(size_t)(unsigned)id * 128u + 64u * half + lane]; };
        auto park_tab = [&](int slot, int half, uint4 v) { reinterpret_cast<uint4*>(LT + 256 * slot)[64 * half + lane] = v; };
        auto load_hdr = [&](int p) { return T.hdr[min(p, npos - 1)]; };
        // running ring offset: position q continues the rows of q - 1, or starts a new run at the next line boundary
        int prev_r0 = 0, prev_nrow = 0;
        unsigned prev_rho = 0;
        auto entry = [&](int4 hq, int slot, bool changed, bool first) {
            const int r0 = rfl(hq.x), nrow = rfl(hq.y);
            const unsigned endp = prev_rho + (unsigned)(SS * prev_nrow);
            const bool cont = !first && r0 == prev_r0 + prev_nrow;
            const unsigned rho = first ? head_of(r0) : (cont ? endp : ((endp + 15u) & ~15u) + head_of(r0));
            prev_r0 = r0; prev_nrow = nrow; prev_rho = rho;
            i32x4 o;
            o.x = r0; o.y = nrow; o.z = (rfl(hq.z) & 1) | (slot << 1) | (changed ? 4 : 0) | (cont ? 8 : 0); o.w = (int)rho;
            return o;
        };
        int4 hq0 = load_hdr(p_begin), hq1 = load_hdr(p_begin + 1);
        int slot_cur = 0;                 // table slot of the newest published position
        int id_prev = rfl(hq0.z) >> 8;
        int chgA = -1, chgB = 0;          // positions (relative) of the last two table changes
        {
            const uint4 t0 = load_tab(id_prev, 0), t1 = load_tab(id_prev, 1);
            park_tab(0, 0, t0); park_tab(0, 1, t1);
            const int id1 = rfl(hq1.z) >> 8;
            const bool ch1 = id1 != id_prev;
            if (ch1) { const uint4 u0 = load_tab(id1, 0), u1 = load_tab(id1, 1); park_tab(1, 0, u0); park_tab(1, 1, u1); slot_cur = 1; chgA = 0; chgB = 1; }
            const i32x4 e0 = entry(hq0, 0, true, true), e1 = entry(hq1, slot_cur, ch1, false);
            if (lane == 0) { HDR[0] = e0; HDR[1] = e1; }
            id_prev = id1;
        }
        f64x2 piece[DEPTH][ROUNDS];
        int e_nxt[DEPTH][ROUNDS];
        int4 h_nxt[DEPTH];
        uint4 tab0 = {0, 0, 0, 0}, tab1 = {0, 0, 0, 0};
        bool tab_pending = false;         // tab0 / tab1 hold the lane table of position i + 1
        int slot_pending = 0, tab_need = 0;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) park_piece(0, r, load_piece(load_elem(p_begin, r), r));
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            h_nxt[k] = load_hdr(p_begin + k + 2);
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int e1 = load_elem(p_begin + k + 1, r);
                e_nxt[k][r] = load_elem(p_begin + k + 1 + DEPTH, r);
                piece[k][r] = load_piece(e1, r);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        lds_barrier();  // B0
        if (ablate_arg & (1 << 29)) __builtin_amdgcn_s_setprio(2);   // FENRIS_HIP_AFFINE_PRIO bit 1: the loader wave ahead of the row waves
        // one step of the loader (a macro, not a lambda: the stages must stay in registers); i = relative position
#define AFFINE_RING_LOADER_STEP(k, i)                                                                                          \
        {                                                                                                                     \
            const int need_rows = max((i) + 2 - DJ, tab_pending ? tab_need : 0);                                              \
            tr_mark(0);                                                                                                       \
            for (;;) {                                                                                                        \
                i32x4 r, s;                                                                                                   \
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"                             \
                             : "=&v"(r), "=&v"(s) : "v"(fl_addr), "v"(fl_addr + 16u) : "memory");                             \
                if (min(min(rfl(r.x), rfl(r.y)), min(rfl(r.z), rfl(r.w))) >= need_rows && rfl(s.y) >= (i) - 5) break;         \
                __builtin_amdgcn_s_sleep(1);                                                                                  \
            }                                                                                                                 \
            tr_mark(1);                                                                                                       \
            if (!(DBG && (ablate & 4))) {                                                                                     \
                _Pragma("unroll") for (int r = 0; r < ROUNDS; ++r) {                                                          \
                    park_piece(((i) + 1) & (DJ - 1), r, piece[k][r]);  /* records of i + 1 */                                 \
                    piece[k][r] = load_piece(e_nxt[k][r], r);          /* records of i + 1 + DEPTH */                         \
                    e_nxt[k][r] = load_elem(p_begin + (i) + 1 + 2 * DEPTH, r);                                                \
                }                                                                                                             \
            }                                                                                                                 \
            if (tracing) { lds_drain(); tr_mark(2); }                                                                         \
            if (tab_pending) { park_tab(slot_pending, 0, tab0); park_tab(slot_pending, 1, tab1); tab_pending = false; }       \
            const int id2 = rfl(h_nxt[k].z) >> 8;                                                                             \
            const bool ch2 = id2 != id_prev;                                                                                  \
            const int slot2 = ch2 ? (slot_cur ^ 1) : slot_cur;                                                                \
            const i32x4 e2 = entry(h_nxt[k], slot2, ch2, false);                                                              \
            if (lane == 0) HDR[((i) + 2) & (NH - 1)] = e2;                                                                    \
            if (ch2) {                                                                                                        \
                tab0 = load_tab(id2, 0); tab1 = load_tab(id2, 1); tab_pending = true; slot_pending = slot2;                   \
                tab_need = chgA + 1; chgA = chgB; chgB = (i) + 2;                                                             \
            }                                                                                                                 \
            slot_cur = slot2;                                                                                                 \
            id_prev = id2;                                                                                                    \
            h_nxt[k] = load_hdr(p_begin + (i) + 2 + DEPTH);                                                                   \
            lds_drain();                                                                                                      \
            if (lane == 0) lds_put1(fl_addr + 24u, (i) + 2);                                                                  \
            tr_mark(3);                                                                                                       \
        }
        if (tracing) tr_t = __builtin_readcyclecounter();
        int i0 = 0;
        for (; i0 + DEPTH <= n - 1; i0 += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) AFFINE_RING_LOADER_STEP(k, i0 + k)
        }
#pragma unroll
        for (int k = 0; k < DEPTH - 1; ++k)
            if (i0 + k < n - 1) AFFINE_RING_LOADER_STEP(k, i0 + k)
#undef AFFINE_RING_LOADER_STEP
        tr_report(1);
        return;
    }

    // ---------------------------------------------------------------------------------------------- row waves
    lds_barrier();  // B0
    uint2 lane_cur = {0u, 0u};
    f64x2 gq0[3] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}}, gq1[3] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};   // Laplace: Ghat of the lane's terms
    const unsigned my_flag = fl_addr + 4u * (unsigned)wave;
    if (tracing) tr_t = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        i32x4 fl, hv;
        tr_mark(0);
        for (;;) {
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(fl), "=&v"(hv) : "v"(fl_addr + 16u), "v"(hdr_addr + 16u * (unsigned)(i & (NH - 1))) : "memory");
            if (rfl(fl.z) >= i + 1) break;
            __builtin_amdgcn_s_sleep(1);
        }
        tr_mark(1);
        const int z = rfl(hv.z);
        const unsigned rho = (unsigned)rfl(hv.w), end = rho + (unsigned)(SS * rfl(hv.y));
        {
            int cons = rfl(fl.x);
            while ((int)(end - (unsigned)cons) > ring) {
                __builtin_amdgcn_s_sleep(1);
                cons = rfl(lds_poll4(fl_addr + 16u).x);
            }
        }
        tr_mark(2);
        if (z & 4) {   // the lane table changed with this position
            lane_cur = LT[256 * ((z >> 1) & 1) + tid];
            if constexpr (LAP) {
                const f64x2* q0 = reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(GH) + ((lane_cur.x >> 5) & 127u) * (GW * 8));
                const f64x2* q1 = reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(GH) + ((lane_cur.x >> 17) & 127u) * (GW * 8));
#pragma unroll
                for (int h = 0; h < 3; ++h) { gq0[h] = q0[h]; gq1[h] = q1[h]; }
            }
        }
        const unsigned x = lane_cur.x, y = lane_cur.y;
        const char* js = reinterpret_cast<const char*>(JS + (size_t)(i & (DJ - 1)) * T.us * GW);
        const char* gh = reinterpret_cast<const char*>(GH);
        const unsigned oR0 = (x & 31u) * (GW * 8), oG0 = ((x >> 5) & 127u) * (GW * 8);
        const unsigned oR1 = ((x >> 12) & 31u) * (GW * 8), oG1 = ((x >> 17) & 127u) * (GW * 8);
        const int grp = (int)((x >> 24) & 3u);
        const bool zero_lane = ((x >> 5) & 127u) == 64u && ((x >> 17) & 127u) == 64u;   // gidx 64 twice = a block without a term: zeros (affine_rows.hip)
        const unsigned rb = rho * 8u + (y & 0xffffu);   // ring byte offset (unwrapped) of the lane's block
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
            if (zero_lane) s = 0.0;
            if ((x >> 28) & 1u) *reinterpret_cast<double*>(OUTc + (rb & maskb)) = s;
        } else {
            double H[3][3];
            if (!(DBG && (ablate & 2))) {
                auto load33 = [&](const char* p_, double (&M)[3][3]) {
                    const f64x2* q = reinterpret_cast<const f64x2*>(p_);
                    const f64x2 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4];
                    M[0][0] = v0.x; M[0][1] = v0.y; M[0][2] = v1.x; M[1][0] = v1.y; M[1][1] = v2.x; M[1][2] = v2.y;
                    M[2][0] = v3.x; M[2][1] = v3.y; M[2][2] = v4.x;
                };
                auto term = [&](const char* pr, const char* pg, bool first) {
                    double R[3][3], Gm[3][3], Tm[3][3];
                    load33(pr, R);
                    load33(pg, Gm);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int s_ = 0; s_ < 3; ++s_) Tm[c][s_] = fma(Gm[c][2], R[2][s_], fma(Gm[c][1], R[1][s_], Gm[c][0] * R[0][s_]));
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int s_ = 0; s_ < 3; ++s_) {
                            double h = first ? R[0][r] * Tm[0][s_] : fma(R[0][r], Tm[0][s_], H[r][s_]);
                            h = fma(R[1][r], Tm[1][s_], h);
                            H[r][s_] = fma(R[2][r], Tm[2][s_], h);
                        }
                };
                term(js + oR0, gh + oG0, true);
                term(js + oR1, gh + oG1, false);
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) H[r][s_] = (double)(x + 3 * r + s_);
            }
            if (grp >= 1) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) H[r][s_] += dpp_quad_full<0xB1>(H[r][s_]);
            }
            if (grp >= 2) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) H[r][s_] += dpp_quad_full<0x4E>(H[r][s_]);
            }
            if (zero_lane) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) H[i][s_] = 0.0;
            }
            if ((x >> 28) & 1u) {
                const bool tr = (x >> 26) & 1u, dg = (x >> 27) & 1u;
                const double mu_tr = a.mu * (H[0][0] + H[1][1] + H[2][2]);
                const double mpl = a.mu + a.lambda;
                double v[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r) v[r][r] = fma(mpl, H[r][r], mu_tr);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int j = r + 1; j < 3; ++j) {
                        const double up = fma(a.mu, H[j][r], a.lambda * H[r][j]);   // (r, j)
                        const double lw = fma(a.mu, H[r][j], a.lambda * H[j][r]);   // (j, r)
                        v[r][j] = tr ? lw : up;
                        v[j][r] = (tr || dg) ? up : lw;
                    }
                const unsigned rs = y >> 16;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const unsigned o = rb + (unsigned)r * rs;
#pragma unroll
                    for (int k = 0; k < 3; ++k) *reinterpret_cast<double*>(OUTc + ((o + 8u * k) & maskb)) = v[r][k];
                }
            }
        }
        tr_mark(0);
        lds_drain();
        if ((tid & 63) == 0) lds_put1(my_flag, i + 1);
        tr_mark(3);
    }
    if (wave == 0) tr_report(0);
}

template <int OP, int DEPTH>
static auto affine_ring_pick(bool ow, bool dbg) -> void (*)(const KArgs, const AffineRowTables, int, int) {
    if (dbg) return k_affine_ring<OP, true, true, DEPTH>;
    return ow ? k_affine_ring<OP, true, false, DEPTH> : k_affine_ring<OP, false, false, DEPTH>;
}
template <int OP>
static auto affine_ring_pick_depth(int depth, bool ow, bool dbg) -> void (*)(const KArgs, const AffineRowTables, int, int) {
    // positions the loader's requests run ahead of their parking (registers): 4 and beyond spill at five waves per SIMD
    return depth <= 2 ? affine_ring_pick<OP, 2>(ow, dbg) : affine_ring_pick<OP, 3>(ow, dbg);
}

hipError_t affine_ring_launch(int op, int ring, int depth, int grid, size_t lds_bytes, hipStream_t stream, const KArgs& a, const AffineRowTables& T,
                              int ablate) {
    const bool ow = a.overwrite != 0, dbg = (ablate & 0xffff) != 0 || a.trace != nullptr;
    void (*kern)(const KArgs, const AffineRowTables, int, int) =
        op == FH_LAPLACE ? affine_ring_pick_depth<FH_LAPLACE>(depth, ow, dbg) : affine_ring_pick_depth<FH_LINEAR_ELASTIC>(depth, ow, dbg);
    if (lds_bytes > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(384), lds_bytes, stream, a, T, ring, ablate);
    return hipGetLastError();
}

}  // namespace fenris_hip
