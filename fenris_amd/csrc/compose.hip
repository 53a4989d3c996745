// Composition of element assemblers behind the C ABI (src/assembly/local.rs:152-340: AggregateElementAssembler,
// MapElementNodes, TransformElement{Matrix,Vector} for the closed family the device can express -- a scale factor): what a
// context assembled in ITS node numbering is added, scaled, into a matrix / vector over another node index space,
//     dst(map[i] s + r, map[j] s + c) += scale * src(i s + r, j s + c),      dst(map[i] s + r) += scale * src(i s + r).
// Several bodies (meshes, operators, tables, element kinds) thus meet in one global K: every body runs its own fastest
// kernels, this pass merges.  One thread per node-level entry of the source pattern; the destination column is found by
// binary search in the destination row (ascending, like every pattern of assemble_pattern, global.rs:100).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/fenris_hip.h"
#include "group_internal.hpp"

namespace {

template <int S>
__global__ void __launch_bounds__(256) k_add_mapped_matrix(int num_nodes, const unsigned* noff, const unsigned* ncols, const double* src,
                                                           const unsigned long long* node_map, double scale,
                                                           unsigned long long dst_nodes, const unsigned long long* dro,
                                                           const unsigned long long* dci, double* dst, int* missing) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread per (source node, local row r)
    if (t >= (long long)num_nodes * S) return;
    const int i = (int)(t / S), r = (int)(t % S);
    const unsigned r0 = noff[i], cnt = noff[i + 1] - r0;
    if (cnt == 0) return;
    const unsigned long long I = node_map ? node_map[i] : (unsigned long long)i;
    if (I >= dst_nodes) { atomicAdd(missing, 1); return; }
    const unsigned long long row = I * S + r, b = dro[row], e = dro[row + 1];
    const double* srow = src + (size_t)S * S * r0 + (size_t)r * S * cnt;
    for (unsigned k = 0; k < cnt; ++k) {
        const unsigned j = ncols[r0 + k];
        const unsigned long long col0 = (node_map ? node_map[j] : (unsigned long long)j) * S;
        unsigned long long lo = b, hi = e;   // first index with dci >= col0
        while (lo < hi) {
            const unsigned long long mid = (lo + hi) >> 1;
            if (dci[mid] < col0) lo = mid + 1; else hi = mid;
        }
        if (lo + S > e || dci[lo] != col0 || dci[lo + S - 1] != col0 + S - 1) { atomicAdd(missing, 1); continue; }
#pragma unroll
        for (int c = 0; c < S; ++c) unsafeAtomicAdd(dst + lo + c, scale * srow[(size_t)S * k + c]);
    }
}

template <int S>
__global__ void __launch_bounds__(256) k_add_mapped_vector(int num_nodes, const double* src, const unsigned long long* node_map,
                                                           double scale, unsigned long long dst_nodes, double* dst, int* missing) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)num_nodes * S) return;
    const int i = (int)(t / S), r = (int)(t % S);
    const unsigned long long I = node_map ? node_map[i] : (unsigned long long)i;
    if (I >= dst_nodes) { atomicAdd(missing, 1); return; }
    const double v = scale * src[t];
    if (v != 0.0) unsafeAtomicAdd(dst + I * S + r, v);
}

int check_missing(fh_ctx* c, int* flag_dev, hipStream_t stream, const char* who) {
    int h = 0;
    if (hipMemcpyAsync(&h, flag_dev, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipFree(flag_dev);
        return fh_internal_fail(c, FH_HIP_ERROR, std::string(who) + ": status read-back failed");
    }
    (void)hipFree(flag_dev);
    // add_element_row_to_csr_row panics on a missing column (global.rs:531-533); here an error with the count
    if (h) return fh_internal_fail(c, FH_BAD_ARGUMENT, std::string(who) + ": " + std::to_string(h) +
                                                         " mapped entries have no place in the destination (node out of range or column missing)");
    return FH_OK;
}

}  // namespace

extern "C" {

int fh_add_mapped_matrix_dev(fh_ctx* c, const double* src_values_dev, const uint64_t* node_map_dev, double scale, uint64_t dst_num_nodes,
                             const uint64_t* dst_row_offsets_dev, const uint64_t* dst_col_indices_dev, double* dst_values_dev) {
    if (!c) return FH_BAD_ARGUMENT;
    const unsigned *noff = nullptr, *ncols = nullptr;
    uint64_t N = 0;
    int S = 0;
    if (!fh_internal_pattern(c, &noff, &ncols, &N, &S)) return fh_internal_fail(c, FH_INVALID_STATE, "fh_add_mapped_matrix: call fh_pattern first");
    if (!src_values_dev || !dst_row_offsets_dev || !dst_col_indices_dev || !dst_values_dev)
        return fh_internal_fail(c, FH_BAD_ARGUMENT, "fh_add_mapped_matrix: null argument");
    if (N == 0) return FH_OK;
    DevGuardExt dev_guard_(fh_internal_device(c));
    hipStream_t stream = fh_internal_stream(c);
    int* flag = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&flag), sizeof(int)) != hipSuccess || hipMemsetAsync(flag, 0, sizeof(int), stream) != hipSuccess)
        return fh_internal_fail(c, FH_HIP_ERROR, "fh_add_mapped_matrix: allocation failed");
    const unsigned grid = (unsigned)((N * (uint64_t)S + 255) / 256);
    const auto* map = reinterpret_cast<const unsigned long long*>(node_map_dev);
    const auto* dro = reinterpret_cast<const unsigned long long*>(dst_row_offsets_dev);
    const auto* dci = reinterpret_cast<const unsigned long long*>(dst_col_indices_dev);
#define LAUNCH(SV) hipLaunchKernelGGL(k_add_mapped_matrix<SV>, dim3(grid), dim3(256), 0, stream, (int)N, noff, ncols, src_values_dev, map, scale, \
                                      (unsigned long long)dst_num_nodes, dro, dci, dst_values_dev, flag)
    if (S == 1) LAUNCH(1); else if (S == 2) LAUNCH(2); else LAUNCH(3);
#undef LAUNCH
    if (hipGetLastError() != hipSuccess) { (void)hipFree(flag); return fh_internal_fail(c, FH_HIP_ERROR, "fh_add_mapped_matrix: launch failed"); }
    return check_missing(c, flag, stream, "fh_add_mapped_matrix");
}

int fh_add_mapped_vector_sdim_dev(fh_ctx* c, const double* src_dev, const uint64_t* node_map_dev, double scale, int solution_dim,
                                  uint64_t dst_num_nodes, double* dst_dev) {
    if (!c) return FH_BAD_ARGUMENT;
    uint64_t N = 0;
    int S = 0;
    if (solution_dim > 0) {   // a source vector: the context holds a mesh but no operator, the source gives the dimension
        if (!fh_internal_num_nodes(c, &N)) return fh_internal_fail(c, FH_INVALID_STATE, "fh_add_mapped_vector: set the mesh first");
        S = solution_dim;
    } else if (!fh_internal_sizes(c, &N, &S)) {
        return fh_internal_fail(c, FH_INVALID_STATE, "fh_add_mapped_vector: set mesh and operator first");
    }
    if (S < 1 || S > 3) return fh_internal_fail(c, FH_BAD_ARGUMENT, "fh_add_mapped_vector: solution dimension must be 1, 2 or 3");
    if (!src_dev || !dst_dev) return fh_internal_fail(c, FH_BAD_ARGUMENT, "fh_add_mapped_vector: null argument");
    if (N == 0) return FH_OK;
    DevGuardExt dev_guard_(fh_internal_device(c));
    hipStream_t stream = fh_internal_stream(c);
    int* flag = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&flag), sizeof(int)) != hipSuccess || hipMemsetAsync(flag, 0, sizeof(int), stream) != hipSuccess)
        return fh_internal_fail(c, FH_HIP_ERROR, "fh_add_mapped_vector: allocation failed");
    const unsigned grid = (unsigned)((N * (uint64_t)S + 255) / 256);
    const auto* map = reinterpret_cast<const unsigned long long*>(node_map_dev);
#define LAUNCH(SV) hipLaunchKernelGGL(k_add_mapped_vector<SV>, dim3(grid), dim3(256), 0, stream, (int)N, src_dev, map, scale, \
                                      (unsigned long long)dst_num_nodes, dst_dev, flag)
    if (S == 1) LAUNCH(1); else if (S == 2) LAUNCH(2); else LAUNCH(3);
#undef LAUNCH
    if (hipGetLastError() != hipSuccess) { (void)hipFree(flag); return fh_internal_fail(c, FH_HIP_ERROR, "fh_add_mapped_vector: launch failed"); }
    return check_missing(c, flag, stream, "fh_add_mapped_vector");
}

int fh_add_mapped_vector_dev(fh_ctx* c, const double* src_dev, const uint64_t* node_map_dev, double scale, uint64_t dst_num_nodes,
                             double* dst_dev) {
    return fh_add_mapped_vector_sdim_dev(c, src_dev, node_map_dev, scale, 0, dst_num_nodes, dst_dev);
}

}  // extern "C"
