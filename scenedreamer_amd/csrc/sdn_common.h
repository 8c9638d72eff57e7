// Shared host-side helpers for libsdnative (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/sdnative.h"

namespace sdn {

char *error_buffer();  // thread-local, 512 bytes
int fail(int code, const char *fmt, ...);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SDN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SDN_OK;
}

template <typename T>
static inline T div_up(T a, T b) {
    return (a + b - 1) / b;
}

}  // namespace sdn

#define SDN_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) return sdn::fail(SDN_ERR_INVALID, __VA_ARGS__); \
    } while (0)
