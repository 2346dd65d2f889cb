// vlg_common.h -- error plumbing shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/vlgae_amd.h"

#define VLG_SR_LOG 0
#define VLG_SR_MAX 1

namespace vlg {

char* error_buffer();   // thread-local, defined in vlg_capi.cpp

inline int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// Environment switches (tools/ A-B timing only) are read ONCE per call site: getenv walks the whole environment and is not
// something a launch path should do on every invocation.  Each expansion owns its function-local static.
#define VLG_ENV(name) ([]() -> const char* { static const char* const v_ = getenv(name); return v_; }())

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error((int)e, "%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

}  // namespace vlg
