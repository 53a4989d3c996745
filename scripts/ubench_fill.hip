// Write-bandwidth ceilings on one MI355X: what a kernel that only has to write the nnz * 8 bytes of the Hex8 elasticity 216^3
// matrix (19.7 GB) can reach.  Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_fill.hip -o gpurun_out/ubench_fill
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x2_u __attribute__((ext_vector_type(2), aligned(8)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

// grid-stride, 16 bytes per lane
template <int NT>
__global__ void __launch_bounds__(256) k_fill_stride(f64x2* out, size_t npair, double v) {
    const f64x2 val = {v, v + 1.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npair; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(val, out + i);
        else out[i] = val;
    }
}

// persistent workgroups, each walks contiguous chunks of `chunk` pairs (the row block of a node block: 7 x 243 doubles)
template <int NT, int ALIGN8, int RR = 0>
__global__ void __launch_bounds__(256) k_fill_chunks(double* out, size_t ndbl, int chunk_dbl, double v) {
    const size_t nchunk = (ndbl + chunk_dbl - 1) / chunk_dbl;
    // RR: chunk c belongs to workgroup c % G (the chunks in flight are neighbours in memory) instead of a contiguous range per
    // workgroup (the chunks in flight are range-size apart)
    const size_t c0 = RR ? blockIdx.x : (size_t)blockIdx.x * nchunk / gridDim.x;
    const size_t c1 = RR ? nchunk : (size_t)(blockIdx.x + 1) * nchunk / gridDim.x;
    const size_t step = RR ? gridDim.x : 1;
    const f64x2 val = {v, v + 1.0};
    for (size_t c = c0; c < c1; c += step) {
        double* base = out + c * chunk_dbl;
        const int n = (int)((c + 1 == nchunk) ? (ndbl - c * chunk_dbl) : chunk_dbl);
        const int npair = n >> 1;
        if (ALIGN8) {
            f64x2_u* o2 = reinterpret_cast<f64x2_u*>(base);
            for (int i = threadIdx.x; i < npair; i += 256) {
                if (NT) __builtin_nontemporal_store(val, o2 + i);
                else o2[i] = val;
            }
        } else {
            f64x2* o2 = reinterpret_cast<f64x2*>(base);
            for (int i = threadIdx.x; i < npair; i += 256) o2[i] = val;
        }
        if ((n & 1) && threadIdx.x == 0) base[n - 1] = v;
    }
}

// grid-stride with every 16-byte store displaced by `shift` doubles (misaligned to 16 / 128 bytes)
__global__ void __launch_bounds__(256) k_fill_stride_shift(double* out, size_t npair, int shift, double v) {
    const f64x2 val = {v, v + 1.0};
    f64x2_u* o2 = reinterpret_cast<f64x2_u*>(out + shift);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npair; i += (size_t)gridDim.x * 256) o2[i] = val;
}

// chunks of arbitrary alignment written in pieces aligned to ABSOLUTE 16-byte / 128-byte addresses: lane t of step k takes
// the 16-byte piece q0 + t + 256 k, q0 = first piece of the 128-byte line that holds the chunk's first byte
// MAP 0: chunk c -> workgroup c % G;  1: contiguous range per workgroup;  2: XCD-aware round robin -- at step k the
// workgroups of XCD x (= blockIdx % 8) take the G / 8 consecutive chunks [k G + x G / 8, + G / 8): neighbours in memory are
// written by the same XCD at about the same time, so the partial lines at chunk boundaries merge in that XCD's L2
template <int MAP>
__global__ void __launch_bounds__(256) k_fill_chunks_abs(double* out, size_t ndbl, int chunk_dbl, double v) {
    const size_t nchunk = (ndbl + chunk_dbl - 1) / chunk_dbl;
    const f64x2 val = {v, v + 1.0};
    const size_t G = gridDim.x;
    size_t c_first = blockIdx.x, c_end = nchunk, c_step = G;
    if (MAP == 1) { c_first = blockIdx.x * nchunk / G; c_end = (blockIdx.x + 1) * nchunk / G; c_step = 1; }
    if (MAP == 2) c_first = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;
    for (size_t c = c_first; c < c_end; c += c_step) {
        const size_t d0 = c * chunk_dbl, d1 = (c + 1 == nchunk) ? ndbl : d0 + chunk_dbl;  // doubles [d0, d1)
        const size_t a0 = (size_t)out / 8 + d0, a1 = (size_t)out / 8 + d1;              // absolute double indices
        const size_t q0 = (a0 / 16) * 8;                                                // first 16-byte piece of the line
        const size_t q1 = (a1 + 1) / 2;
        for (size_t q = q0 + threadIdx.x; q < q1; q += 256) {
            const size_t lo = 2 * q, hi = 2 * q + 2;
            double* pd = reinterpret_cast<double*>(q * 16);
            if (lo >= a0 && hi <= a1) *reinterpret_cast<f64x2*>(pd) = val;
            else if (lo >= a0 && lo < a1) pd[0] = v;
            else if (lo + 1 >= a0 && lo + 1 < a1) pd[1] = v;
        }
    }
}

// 8-byte stores, lanes 24 bytes apart (a lane holds a 3 x 3 block: nine stores of one double each)
__global__ void __launch_bounds__(256) k_fill_blocks3(double* out, size_t nnodes, double v) {
    // node block of 243 doubles: 27 lanes x (3 rows x 3 cols); row r of the node is 81 contiguous doubles
    const size_t total = nnodes * 27;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t node = i / 27;
        const int col = (int)(i % 27);
        double* base = out + node * 243 + 3 * col;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) base[r * 81 + k] = v + k;
    }
}

// read + write mix: reads `rd` bytes per 16 written (table traffic next to the value stream)
__global__ void __launch_bounds__(256) k_fill_mix(f64x2* out, size_t npair, const f64x2* in, size_t nin, int every, double v) {
    f64x2 acc = {v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npair; i += (size_t)gridDim.x * 256) {
        if ((i / 256) % every == 0) {
            const f64x2 t = in[(i / every) % nin];
            acc.x += t.x;
            acc.y += t.y;
        }
        out[i] = acc;
    }
}


// the store wave of k_affine_rows (affine_rows.hip) alone: workgroups of ONE wave, each walks a contiguous range of chunks and
// writes whole 128-byte lines, 16 bytes per lane, four stores in flight per trip; `spin` dummy VALU iterations between chunks
// stand for the time the row waves take (0 = stores back to back)
__global__ void __launch_bounds__(64) k_fill_store_wave(double* out, size_t ndbl, int chunk_dbl, int spin, double v) {
    const size_t nline = ndbl / 16;                         // whole lines only
    const size_t lines_per_chunk = (size_t)chunk_dbl / 16;  // 13 lines + carry ~ 1664 doubles
    const size_t nchunk = nline / lines_per_chunk;
    const size_t c0 = (size_t)blockIdx.x * nchunk / gridDim.x, c1 = (size_t)(blockIdx.x + 1) * nchunk / gridDim.x;
    f64x2 val = {v, v + 1.0};
    const int npiece = (int)lines_per_chunk * 8;
    for (size_t c = c0; c < c1; ++c) {
        f64x2* o2 = reinterpret_cast<f64x2*>(out + c * lines_per_chunk * 16);
        int k = threadIdx.x;
        for (; k + 192 < npiece; k += 256) { o2[k] = val; o2[k + 64] = val; o2[k + 128] = val; o2[k + 192] = val; }
        for (; k < npiece; k += 64) o2[k] = val;
        for (int i = 0; i < spin; ++i) val.x = fma(val.x, 1.0000001, 1e-9);
    }
}

int main(int argc, char** argv) {
    const size_t ndbl = (argc > 1) ? std::strtoull(argv[1], nullptr, 10) : 2460235041ull;  // nnz of Hex8 elasticity 216^3
    const int reps = 5;
    double* buf = nullptr;
    CHECK(hipMalloc((void**)&buf, (ndbl + 2) * 8));
    f64x2* rd = nullptr;
    const size_t nin = (size_t)1 << 28;  // 4 GB read stream
    CHECK(hipMalloc((void**)&rd, nin * 16));
    CHECK(hipMemset(rd, 0, nin * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& launch, double bytes) {
        launch();
        CHECK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps; ++r) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            sum += ms;
        }
        std::printf("{\"variant\": \"%s\", \"ms_best\": %.3f, \"ms_avg\": %.3f, \"GBps_best\": %.1f}\n", name, best, sum / reps,
                    bytes / best * 1e-6);
        std::fflush(stdout);
    };
    const double B = (double)ndbl * 8.0;
    const size_t npair = ndbl / 2;
    for (int g : {2048, 4096, 16384}) {
        char nm[64];
        std::snprintf(nm, sizeof nm, "stride16_grid%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL(k_fill_stride<0>, dim3(g), dim3(256), 0, 0, (f64x2*)buf, npair, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "stride16_nt_grid%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL(k_fill_stride<1>, dim3(g), dim3(256), 0, 0, (f64x2*)buf, npair, 1.0); }, B);
    }
    timeit("memset", [&] { CHECK(hipMemsetAsync(buf, 0, ndbl * 8, 0)); }, B);
    for (int wg : {2, 4, 8}) {
        char nm[64];
        std::snprintf(nm, sizeof nm, "chunks1701_a8_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_fill_chunks<0, 1>), dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1701, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "chunks1701_a8_nt_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_fill_chunks<1, 1>), dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1701, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "chunks1702_a16_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_fill_chunks<0, 0>), dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1702, 1.0); }, B);
    }
    for (int wg : {2, 3, 4, 8}) {
        char nm[64];
        std::snprintf(nm, sizeof nm, "rr_chunks1701_a8_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_fill_chunks<0, 1, 1>), dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1701, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "rr_chunks1701_a8_nt_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_fill_chunks<1, 1, 1>), dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1701, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "rr_chunks1702_a16_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_fill_chunks<0, 0, 1>), dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1702, 1.0); }, B);
    }
    timeit("stride16_shift1_grid16384", [&] { hipLaunchKernelGGL(k_fill_stride_shift, dim3(16384), dim3(256), 0, 0, buf, npair - 1, 1, 1.0); }, B);
    timeit("stride16_shift2_grid16384", [&] { hipLaunchKernelGGL(k_fill_stride_shift, dim3(16384), dim3(256), 0, 0, buf, npair - 1, 2, 1.0); }, B);
    for (int wg : {2, 4, 8}) {
        char nm[64];
        std::snprintf(nm, sizeof nm, "rr_chunks1664_a128_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_fill_chunks<0, 0, 1>), dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1664, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "rr_chunks2048_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_fill_chunks<0, 0, 1>), dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 2048, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "rr_chunks1701_abs_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL(k_fill_chunks_abs<0>, dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1701, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "contig_chunks1701_abs_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL(k_fill_chunks_abs<1>, dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1701, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "xcdrr_chunks1701_abs_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL(k_fill_chunks_abs<2>, dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1701, 1.0); }, B);
        std::snprintf(nm, sizeof nm, "xcdrr_chunks1664_abs_%dwg", wg);
        timeit(nm, [&] { hipLaunchKernelGGL(k_fill_chunks_abs<2>, dim3(256 * wg), dim3(256), 0, 0, buf, ndbl, 1664, 1.0); }, B);
    }
    timeit("blocks3x3_8B", [&] { hipLaunchKernelGGL(k_fill_blocks3, dim3(8192), dim3(256), 0, 0, buf, ndbl / 243, 1.0); }, (double)(ndbl / 243) * 243 * 8);
    for (int every : {4, 8}) {
        char nm[64];
        std::snprintf(nm, sizeof nm, "mix_read1_per_%d", every);
        timeit(nm, [&] { hipLaunchKernelGGL(k_fill_mix, dim3(4096), dim3(256), 0, 0, (f64x2*)buf, npair, rd, nin, every, 1.0); },
               B * (1.0 + 1.0 / every));
    }
    for (int wg : {2, 3, 4, 6, 8})
        for (int spin : {0, 200}) {
            char nm[64];
            std::snprintf(nm, sizeof nm, "store_wave_%dwg_spin%d", wg, spin);
            timeit(nm, [&] { hipLaunchKernelGGL(k_fill_store_wave, dim3(256 * wg), dim3(64), 0, 0, buf, ndbl, 1701, spin, 1.0); }, B);
        }
    return 0;
}
