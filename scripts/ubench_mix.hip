// Write stream of k_affine_rows next to a read stream: how much does the value stream (19.7 GB, whole 128-byte lines, one store
// wave per workgroup, three workgroups per CU) lose when another wave of the same workgroup reads R bytes per chunk from memory
// that no cache holds?  Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_mix.hip -o gpurun_out/ubench_mix
//   variant  rd<k>_<sync>: k rounds of 64 x 16-byte loads per chunk of 106 lines (1696 doubles); sync = bar (one s_barrier per chunk,
//   like the kernel before round 3), free (no synchronisation: the reader runs at its own pace), flag (reader ahead by at most
//   4 chunks through an LDS counter)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double f64x2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                         \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

template <int SYNC>  // 0 free, 1 barrier per chunk, 2 LDS counter (reader at most 4 chunks ahead)
__global__ void __launch_bounds__(128) k_mix(double* out, size_t ndbl, const f64x2* in, size_t nin, int rounds, int stride_pieces, double v,
                                             double* sink) {
    __shared__ volatile int done_store;
    const size_t lines_per_chunk = 106;  // 1696 doubles: a position of seven nodes
    const size_t nchunk = (ndbl / 16) / lines_per_chunk;
    const size_t c0 = (size_t)blockIdx.x * nchunk / gridDim.x, c1 = (size_t)(blockIdx.x + 1) * nchunk / gridDim.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) done_store = 0;
    __syncthreads();
    if (wave == 0) {
        const f64x2 val = {v, v + 1.0};
        const int npiece = (int)lines_per_chunk * 8;
        for (size_t c = c0; c < c1; ++c) {
            f64x2* o2 = reinterpret_cast<f64x2*>(out + c * lines_per_chunk * 16);
            int k = lane;
            for (; k + 192 < npiece; k += 256) { o2[k] = val; o2[k + 64] = val; o2[k + 128] = val; o2[k + 192] = val; }
            for (; k < npiece; k += 64) o2[k] = val;
            if (SYNC == 1) __builtin_amdgcn_s_barrier();
            if (SYNC == 2 && lane == 0) done_store = (int)(c - c0) + 1;
        }
    } else {
        f64x2 acc = {0.0, 0.0};
        // the reader's chunk c reads `rounds` KB starting at a pseudo-random place (records of 80 bytes in runs of 640)
        for (size_t c = c0; c < c1; ++c) {
            if (SYNC == 2) {
                while ((int)(c - c0) - done_store > 4) __builtin_amdgcn_s_sleep(2);
            }
            const size_t base = ((c * 2654435761ull) % (nin / 1024)) * 1024;
            for (int r = 0; r < rounds; ++r) {
                const f64x2 t = in[(base + (size_t)(r * 64 + lane) * (size_t)stride_pieces) % nin];
                acc.x += t.x;
                acc.y += t.y;
            }
            if (SYNC == 1) __builtin_amdgcn_s_barrier();
        }
        if (acc.x == 123.456) sink[0] = acc.y;
    }
}

int main(int argc, char** argv) {
    const size_t ndbl = 2460235041ull;  // nnz of Hex8 elasticity 216^3
    const int reps = 5;
    double* buf = nullptr;
    CHECK(hipMalloc((void**)&buf, (ndbl + 2) * 8));
    f64x2* rd = nullptr;
    const size_t nin = (size_t)1 << 28;  // 4 GB read stream
    CHECK(hipMalloc((void**)&rd, nin * 16));
    CHECK(hipMemset(rd, 0, nin * 16));
    double* sink = nullptr;
    CHECK(hipMalloc((void**)&sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t nchunk = (ndbl / 16) / 106;
    auto timeit = [&](const char* name, auto&& launch, double wbytes, double rbytes) {
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
        std::printf("{\"variant\": \"%s\", \"ms_best\": %.3f, \"ms_avg\": %.3f, \"write_GBps\": %.1f, \"read_GB\": %.2f, \"total_GBps\": %.1f}\n", name, best,
                    sum / reps, wbytes / best * 1e-6, rbytes * 1e-9, (wbytes + rbytes) / best * 1e-6);
        std::fflush(stdout);
    };
    const double B = (double)ndbl * 8.0;
    for (int wg : {3}) {
        for (int stride : {1, 5}) {
            for (int rounds : {0, 1, 2, 3, 5, 8}) {
                char nm[96];
                const double rb = (double)nchunk * rounds * 1024.0;
                std::snprintf(nm, sizeof nm, "wg%d_rd%d_s%d_bar", wg, rounds, stride);
                timeit(nm, [&] { hipLaunchKernelGGL(k_mix<1>, dim3(256 * wg), dim3(128), 0, 0, buf, ndbl, rd, nin, rounds, stride, 1.0, sink); }, B, rb);
                std::snprintf(nm, sizeof nm, "wg%d_rd%d_s%d_free", wg, rounds, stride);
                timeit(nm, [&] { hipLaunchKernelGGL(k_mix<0>, dim3(256 * wg), dim3(128), 0, 0, buf, ndbl, rd, nin, rounds, stride, 1.0, sink); }, B, rb);
                std::snprintf(nm, sizeof nm, "wg%d_rd%d_s%d_flag", wg, rounds, stride);
                timeit(nm, [&] { hipLaunchKernelGGL(k_mix<2>, dim3(256 * wg), dim3(128), 0, 0, buf, ndbl, rd, nin, rounds, stride, 1.0, sink); }, B, rb);
            }
        }
    }
    return 0;
}
