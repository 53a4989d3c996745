// Semantics check for global_load_lds_dwordx3 on gfx950: lane l copies 12 bytes from its own global address to LDS base + 12 l ?
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/dma3_test.hip -o scripts/bin/dma3_test && scripts/bin/dma3_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma12(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void k(const unsigned* src, const int* idx, unsigned* out) {
    __shared__ unsigned buf[64 * 3 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 3 + 64; i += 64) buf[i] = 0xdeadbeefu;
    __syncthreads();
    if (lane < 40) dma12(src + 3 * idx[lane], __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)buf));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 64 * 3 + 64; i += 64) out[i] = buf[i];
}
int main() {
    std::vector<unsigned> h(3000);
    for (int i = 0; i < 3000; ++i) h[i] = i;
    std::vector<int> idx(64);
    for (int i = 0; i < 64; ++i) idx[i] = (i * 37 + 5) % 999;
    unsigned *d, *o; int* di;
    hipMalloc(&d, 12000); hipMalloc(&o, 4 * 256); hipMalloc(&di, 256);
    hipMemcpy(d, h.data(), 12000, hipMemcpyHostToDevice);
    hipMemcpy(di, idx.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, di, o);
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 40; ++l) for (int c = 0; c < 3; ++c) if (r[3 * l + c] != (unsigned)(3 * idx[l] + c)) ++bad;
    for (int i = 120; i < 256; ++i) if (r[i] != 0xdeadbeefu) ++bad;
    printf("dwordx3 lds-dma: lane stride 12 bytes, inactive lanes skipped: %s (bad=%d) first words: %u %u %u %u %u %u\n", bad ? "NO" : "yes", bad, r[0], r[1], r[2], r[3], r[4], r[5]);
    return bad != 0;
}
