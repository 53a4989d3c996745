// Register layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by experiment: one-hot A at lane la, one-hot B at lane lb -> which lane of D is 1?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f64_4x4_layout.hip -o scripts/bin/mfma_f64_4x4_layout
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) k_probe(int* out) {
    const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
    const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    if (d != 0.0) out[blockIdx.x] = lane + 64 * (int)d;   // (at most one lane: a product of two one-hots)
}

int main() {
    int* out;
    hipMalloc(&out, sizeof(int) * 4096);
    hipMemset(out, 0xff, sizeof(int) * 4096);
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, out);
    std::vector<int> h(4096);
    hipMemcpy(h.data(), out, sizeof(int) * 4096, hipMemcpyDeviceToHost);
    // for every A lane: the B lanes it meets and the D lanes that result
    for (int la = 0; la < 64; ++la) {
        std::printf("A lane %2d meets B lanes -> D lane:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (h[la * 64 + lb] >= 0) std::printf("  %2d->%2d", lb, h[la * 64 + lb] % 64);
        std::printf("\n");
    }
    return 0;
}
