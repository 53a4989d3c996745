// Which CU does bit i of a hipExtStreamCreateWithCUMask mask enable on this device?  One stream per bit, one small kernel each; prints
// bit -> (XCC_ID, HW_ID fields).   hipcc --offload-arch=gfx950 -O2 scripts/ubench/cu_mask_map.hip -o scripts/bin/cu_mask_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}
int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int words = (cus + 31) / 32;
    unsigned* d;
    hipMalloc(&d, 8 * 64);
    std::vector<unsigned> h(128);
    for (int bit = 0; bit < cus; ++bit) {
        std::vector<uint32_t> m(words, 0u);
        m[bit / 32] = 1u << (bit % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, words, m.data()) != hipSuccess) { printf("bit %d: stream creation failed\n", bit); continue; }
        hipMemsetAsync(d, 0xff, 8 * 64, s);
        hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, 8 * 32, hipMemcpyDeviceToHost);
        // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx940: se 3 bits)
        printf("bit %3d:", bit);
        unsigned last = ~0u;
        for (int b = 0; b < 16; ++b) {
            const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
            const unsigned key = (xcc << 16) | (hw & 0xff00);
            if (key != last) printf("  xcc %u se %u sh %u cu %2u", xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
            last = key;
        }
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
