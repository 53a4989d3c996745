// What group.hip needs from the context (defined in engine.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <string>

struct fh_ctx;
int fh_internal_fail(fh_ctx* c, int code, const std::string& msg);
int fh_internal_device(const fh_ctx* c);
hipStream_t fh_internal_stream(const fh_ctx* c);
// node-level pattern of the context (device arrays); false when no pattern has been built
bool fh_internal_pattern(const fh_ctx* c, const unsigned** noff, const unsigned** ncols, uint64_t* num_nodes, int* solution_dim);
// generation of that pattern (0: none): changes whenever the arrays above are rebuilt -- holders of the pointers compare it before use
unsigned long long fh_internal_pattern_gen(const fh_ctx* c);
bool fh_internal_sizes(const fh_ctx* c, uint64_t* num_nodes, int* solution_dim);
bool fh_internal_num_nodes(const fh_ctx* c, uint64_t* num_nodes);   // false without a mesh

// the context's device for the duration of a call; the calling thread's current device is restored on return
struct DevGuardExt {
    int prev = -1;
    explicit DevGuardExt(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != dev) { prev = cur; (void)hipSetDevice(dev); }
    }
    ~DevGuardExt() { if (prev >= 0) (void)hipSetDevice(prev); }
    DevGuardExt(const DevGuardExt&) = delete;
    DevGuardExt& operator=(const DevGuardExt&) = delete;
};
