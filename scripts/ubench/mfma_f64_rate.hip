// fp64 matrix instructions on gfx950: cycles per instruction and sustained flop rate of v_mfma_f64_16x16x4_f64 (2 048 flop) and of
// v_mfma_f64_4x4x4_4b_f64 (four 4 x 4 x 4 blocks: 512 flop) -- the question of the round-4 review for C4: would 28-padding of the 27 x 27
// operands (7 x 7 blocks of 4) cost the same matrix-core time per flop as the 32-padding of the 16 x 16 tiles?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f64_rate.hip -o scripts/bin/mfma_f64_rate && scripts/bin/mfma_f64_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double f64x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                         \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

// NACC independent accumulators, ITER trips of NACC instructions each
template <int KIND, int NACC>
__global__ void __launch_bounds__(256) k_rate(double* out, unsigned long long* cycles, int iters, double a0, double b0) {
    f64x4 acc16[NACC];
    double acc4[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc16[i] = f64x4{0, 0, 0, 0}; acc4[i] = 0.0; }
    const double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += (KIND == 0) ? (acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3]) : acc4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND, int NACC>
static void run(const char* name, int wgs_per_cu, double flop_per_inst) {
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = cus * wgs_per_cu, iters = 20000;
    double* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, sizeof(double) * grid * 256));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * grid));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<KIND, NACC>), dim3(grid), dim3(256), 0, 0, out, cyc, 100, 1.0, 2.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<KIND, NACC>), dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0, 2.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c0 = 0;
    CHECK(hipMemcpy(&c0, cyc, sizeof c0, hipMemcpyDeviceToHost));
    const double insts = (double)grid * 4 * iters * NACC;   // wavefront instructions
    std::printf("%-28s %d accumulators, %d wavefront(s) per SIMD: %7.2f ms  %6.1f ns per instruction and SIMD  %7.2f TFLOP/s\n", name, NACC,
                wgs_per_cu, ms, ms * 1e6 / ((double)iters * NACC * wgs_per_cu), insts * flop_per_inst / (ms * 1e-3) / 1e12);
    CHECK(hipFree(out));
    CHECK(hipFree(cyc));
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<0, 1>("v_mfma_f64_16x16x4_f64", w, 2048.0);
        run<0, 2>("v_mfma_f64_16x16x4_f64", w, 2048.0);
        run<0, 4>("v_mfma_f64_16x16x4_f64", w, 2048.0);
        run<0, 7>("v_mfma_f64_16x16x4_f64", w, 2048.0);
        run<1, 1>("v_mfma_f64_4x4x4_4b_f64", w, 512.0);
        run<1, 4>("v_mfma_f64_4x4x4_4b_f64", w, 512.0);
        run<1, 8>("v_mfma_f64_4x4x4_4b_f64", w, 512.0);
        run<1, 14>("v_mfma_f64_4x4x4_4b_f64", w, 512.0);
    }
    return 0;
}
