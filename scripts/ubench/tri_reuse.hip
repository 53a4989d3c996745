// What would an element-block second pass of C4 cost in memory time?  (Round-4 review, option (a): store only the node-block upper triangle of
// K_e -- 3 402 doubles instead of 6 561 -- and let the workgroup that owns an element's newly numbered nodes read the triangles of the SEVEN
// elements those nodes touch: e, e + 1, e + nx, e + nx + 1, e + nx ny, e + nx ny + 1, e + nx ny + nx for the 50 x 50 x 80 box.)  This kernel
// has exactly that traffic and nothing else: per element the k neighbour triangles are read whole (27 216 bytes each, coalesced 16-byte
// loads), summed, and 37 128 bytes of "values" are written; elements are dealt so that each XCD walks one contiguous eighth of them
// (neighbours in x and y meet in that XCD's L2; the z neighbour is nx ny elements = 68 MB of triangles away: memory-side cache or HBM).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/tri_reuse.hip -o scripts/bin/tri_reuse && scripts/bin/tri_reuse
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

constexpr int TRI = 3402, OUTW = 4642;   // doubles per element: upper block triangle of K_e; its share of the 927 660 969 values (even: 16-byte stores)

template <int K>
__global__ void __launch_bounds__(256) k_tri(const double* tri, double* out, int E, int nx, int nxy, double* sink) {
    const int per_xcd = (E + 7) / 8, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int offs[7] = {0, 1, nx, nx + 1, nxy, nxy + 1, nxy + nx};
    f64x2 acc = {0.0, 0.0};
    for (int i = slot; i < per_xcd; i += nslot) {
        const int e = xcd * per_xcd + i;
        if (e >= E) break;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int en = min(e + offs[k], E - 1);
            const f64x2* src = reinterpret_cast<const f64x2*>(tri + (size_t)en * TRI);
            for (int t = threadIdx.x; t < TRI / 2; t += 256) { const f64x2 v = src[t]; acc.x += v.x; acc.y += v.y; }
        }
        f64x2* dst = reinterpret_cast<f64x2*>(out + (size_t)e * OUTW);
        for (int t = threadIdx.x; t < OUTW / 2; t += 256) dst[t] = acc;
    }
    if (acc.x == 1.2345e300) sink[0] = acc.y;
}

template <int K>
static void run(const double* tri, double* out, int E, int wgs_per_cu, double* sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k_tri<K>), dim3(grid), dim3(256), 0, 0, tri, out, E, 50, 2500, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_tri<K>), dim3(grid), dim3(256), 0, 0, tri, out, E, 50, 2500, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double uniq = (double)E * (TRI + OUTW) * 8.0, req = (double)E * ((double)K * TRI + OUTW) * 8.0;
    std::printf("%d triangle(s) per element, %d workgroups per CU: %6.3f ms   unique bytes %5.2f GB -> %5.2f TB/s   requested %5.2f GB -> %5.2f TB/s\n", K,
                wgs_per_cu, ms, uniq / 1e9, uniq / ms / 1e9, req / 1e9, req / ms / 1e9);
}

int main() {
    const int E = 200000;
    double *tri, *out, *sink;
    CHECK(hipMalloc(&tri, sizeof(double) * (size_t)E * TRI));
    CHECK(hipMalloc(&out, sizeof(double) * (size_t)E * OUTW));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(tri, 0, sizeof(double) * (size_t)E * TRI));
    for (int w : {4, 8}) {
        run<1>(tri, out, E, w, sink);
        run<2>(tri, out, E, w, sink);
        run<4>(tri, out, E, w, sink);
        run<7>(tri, out, E, w, sink);
    }
    return 0;
}
