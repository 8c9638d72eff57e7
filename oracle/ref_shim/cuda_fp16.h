// TEST INFRASTRUCTURE: __half / __half2 for the host build of gridencoder.cu (backward, f16 tables).
#ifndef SDN_REF_SHIM_CUDA_FP16_H
#define SDN_REF_SHIM_CUDA_FP16_H
#include "cuda_runtime.h"
#include <c10/util/Half.h>
typedef c10::Half __half;
struct __half2 { __half x, y; };
// One CAS on the 32-bit pair, adding each half in f16 arithmetic (what the device instruction does).
static inline __half2 atomicAdd(__half2* a, __half2 v) {
    uint32_t* p = reinterpret_cast<uint32_t*>(a);
    uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED), nw;
    __half2 o;
    do {
        std::memcpy(&o, &old, 4);
        __half2 n{(__half)((float)o.x + (float)v.x), (__half)((float)o.y + (float)v.y)};
        std::memcpy(&nw, &n, 4);
    } while (!__atomic_compare_exchange_n(p, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return o;
}
#endif
