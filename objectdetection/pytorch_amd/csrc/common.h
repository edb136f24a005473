// Shared host-side helpers for the mi355det C ABI (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "mi355det.h"

namespace mi355 {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MI355_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MI355_OK;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mi355

#define MI355_REQUIRE(cond, ...) \
    do { if (!(cond)) return mi355::fail(MI355_EINVAL, __VA_ARGS__); } while (0)
