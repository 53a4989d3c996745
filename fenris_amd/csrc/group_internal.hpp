// What group.hip needs from the context (defined in engine.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

struct fh_ctx;
int fh_internal_fail(fh_ctx* c, int code, const std::string& msg);
int fh_internal_device(const fh_ctx* c);
hipStream_t fh_internal_stream(const fh_ctx* c);
