// Small dense kernels (2x2 / 3x3 determinant, adjugate, reciprocal square root), hand-issued LDS operations and DPP
// exchanges shared by the assembly kernels.  Only __device__ __forceinline__ functions: safe to include from several
// translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace fenris_hip {

// ------------------------------------------------------------------------------------------ small dense
template <int D> __device__ __forceinline__ double det_small(const double (&m)[D][D]);
template <> __device__ __forceinline__ double det_small<2>(const double (&m)[2][2]) {
    return m[0][0] * m[1][1] - m[1][0] * m[0][1];
}
template <> __device__ __forceinline__ double det_small<3>(const double (&m)[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[0][1] * (m[1][0] * m[2][2] - m[2][0] * m[1][2]) +
           m[0][2] * (m[1][0] * m[2][1] - m[2][0] * m[1][1]);
}
// adjugate times r: the inverse by cofactors / det (what nalgebra's try_inverse does for 2x2 / 3x3) for r = 1 / det
__device__ __forceinline__ void adj_scaled(const double (&m)[2][2], double r, double (&o)[2][2]) {
    o[0][0] = m[1][1] * r;  o[0][1] = -m[0][1] * r;
    o[1][0] = -m[1][0] * r; o[1][1] = m[0][0] * r;
}
__device__ __forceinline__ void adj_scaled(const double (&m)[3][3], double r, double (&o)[3][3]);
template <int D>
__device__ __forceinline__ void inv_small(const double (&m)[D][D], double det, double (&o)[D][D]) {
    adj_scaled(m, 1.0 / det, o);
}
// 1 / sqrt(x) for x > 0 to double precision: v_rsq_f64 seed + two Newton steps (the seed is good to ~2^-26)
__device__ __forceinline__ double rsqrt_newton(double x) {
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double h = fma(-x * y, y, 1.0);  // 1 - x y^2
        y = fma(0.5 * y, h, y);
    }
    return y;
}
__device__ __forceinline__ void adj_scaled(const double (&m)[3][3], double r, double (&o)[3][3]) {
    o[0][0] = (m[1][1] * m[2][2] - m[2][1] * m[1][2]) * r;
    o[0][1] = (m[0][2] * m[2][1] - m[2][2] * m[0][1]) * r;
    o[0][2] = (m[0][1] * m[1][2] - m[1][1] * m[0][2]) * r;
    o[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * r;
    o[1][1] = (m[0][0] * m[2][2] - m[2][0] * m[0][2]) * r;
    o[1][2] = (m[0][2] * m[1][0] - m[1][2] * m[0][0]) * r;
    o[2][0] = (m[1][0] * m[2][1] - m[2][0] * m[1][1]) * r;
    o[2][1] = (m[0][1] * m[2][0] - m[2][1] * m[0][0]) * r;
    o[2][2] = (m[0][0] * m[1][1] - m[1][0] * m[0][1]) * r;
}

// explicit LDS fetch of one double (ds_read_b64).  The generic address of an LDS object carries the LDS byte
// offset in its low 32 bits.  Callers must call lds_wait_all() before using the values.
template <int OFF_BYTES>
__device__ __forceinline__ double lds_read_f64(const double* p) {
    double v;
    const unsigned addr = (unsigned)(unsigned long long)p;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF_BYTES));
    return v;
}
typedef double f64x2 __attribute__((ext_vector_type(2)));
// ds_read_b128 of two adjacent doubles (16-byte aligned): 4 LDS cycles per wave for 16 bytes per lane, and -- unlike
// the 8-byte form -- the rate is reached with one wave per SIMD (MI355X_MICROARCH.md, LDS table)
template <int OFF_BYTES>
__device__ __forceinline__ f64x2 lds_read_f64x2(unsigned addr) {
    f64x2 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF_BYTES));
    return v;
}
template <int OFF_BYTES>
__device__ __forceinline__ double lds_read_f64_at(unsigned addr) {
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF_BYTES));
    return v;
}
template <int D>
__device__ __forceinline__ void lds_read_vec(const double* p, double (&v)[D]) {
    v[0] = lds_read_f64<0>(p);
    v[1] = lds_read_f64<8>(p);
    if (D == 3) v[D - 1] = lds_read_f64<16>(p);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)), which would force the register prefetch of the next blocks to land at every barrier;
// the data exchanged between the waves of the pipelined kernel lives in LDS, so lgkmcnt(0) + s_barrier suffices.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// wait until at most PENDING LDS operations of this wave are outstanding (they complete in order)
template <int PENDING>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PENDING) : "memory");
    __builtin_amdgcn_sched_barrier(0);  // keep consumers below the wait (cdna_hip_programming.md rule 18)
}
__device__ __forceinline__ void lds_wait_all() { lds_wait<0>(); }

template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

// the same exchange when every source lane is known to be active (quad permutes inside a fully active wave, or among lanes
// that take a branch together): no `old` operand to initialise -- two v_mov_b32_dpp instead of four moves and a nop
template <int CTRL>
__device__ __forceinline__ double dpp_quad_full(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

// value of lane ^ 4 (row_shl:4 into the lanes with bit 2 clear, row_shr:4 into the others)
__device__ __forceinline__ double dpp_xor4(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x104, 0xF, 0x5, false);
    lo = __builtin_amdgcn_update_dpp(lo, (int)b, 0x114, 0xF, 0xA, false);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x104, 0xF, 0x5, false);
    hi = __builtin_amdgcn_update_dpp(hi, (int)(b >> 32), 0x114, 0xF, 0xA, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

}  // namespace fenris_hip
