// Would a fused C4 kernel -- element matrices into a small reused ring, the row gather reading them back shortly after (round-4 review, option (b))
// -- keep the dense K_e off HBM?  The TRAFFIC of that design and nothing else: every workgroup, per element, writes a dense block of 52 488 bytes
// into slot (e mod ring) of a ring, reads the block of element e - delay back from the ring (whoever wrote it), and writes its 37 128 bytes of
// "values".  Ring = the whole 10.5 GB is today's two passes in one kernel (every dense byte goes to memory and comes back); a ring of 64 MB - 1 GB
// shows what the memory-side cache absorbs.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/ring_reuse.hip -o scripts/bin/ring_reuse && scripts/bin/ring_reuse
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

constexpr int KE = 6562, OUTW = 4642;   // doubles per element (even: 16-byte pieces)

__global__ void __launch_bounds__(256) k_ring(double* ring, int ring_elems, double* out, int E, int delay, int do_read, double* sink) {
    f64x2 acc = {1.0, 2.0};
    for (int e = blockIdx.x; e < E; e += gridDim.x) {
        f64x2* w = reinterpret_cast<f64x2*>(ring + (size_t)(e % ring_elems) * KE);
        for (int t = threadIdx.x; t < KE / 2; t += 256) w[t] = acc;
        if (do_read) {
            const int er = e >= delay ? e - delay : e;
            const f64x2* r = reinterpret_cast<const f64x2*>(ring + (size_t)(er % ring_elems) * KE);
            for (int t = threadIdx.x; t < KE / 2; t += 256) { const f64x2 v = r[t]; acc.x += v.x * 1e-30; acc.y += v.y * 1e-30; }
        }
        f64x2* dst = reinterpret_cast<f64x2*>(out + (size_t)e * OUTW);
        for (int t = threadIdx.x; t < OUTW / 2; t += 256) dst[t] = acc;
    }
    if (acc.x == 1.2345e300) sink[0] = acc.y;
}

int main() {
    const int E = 200000;
    double *ring, *out, *sink;
    CHECK(hipMalloc(&ring, sizeof(double) * (size_t)E * KE));
    CHECK(hipMalloc(&out, sizeof(double) * (size_t)E * OUTW));
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * 3;
    for (int do_read = 1; do_read >= 0; --do_read)
        for (int ring_mb : {32, 64, 128, 256, 512, 1024, 4096, 10500}) {
            const int ring_elems = (int)std::min<long long>(E, std::max<long long>(grid * 2, (long long)ring_mb * 1000000ll / (KE * 8)));
            const int delay = std::min(ring_elems / 2, 4 * grid);   // read back what was written ~4 rounds of the grid ago (or half a ring)
            hipLaunchKernelGGL(k_ring, dim3(grid), dim3(256), 0, 0, ring, ring_elems, out, E, delay, do_read, sink);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_ring, dim3(grid), dim3(256), 0, 0, ring, ring_elems, out, E, delay, do_read, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 3;
            std::printf("ring %6d MB (%6d elements, read back %5d elements later)%s: %6.3f ms   values alone would be %.2f ms at 4.9 TB/s\n", ring_mb, ring_elems, delay,
                        do_read ? "" : " WRITE ONLY", ms, (double)E * OUTW * 8.0 / 4.9e12 * 1e3);
        }
    return 0;
}
