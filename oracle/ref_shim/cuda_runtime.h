// TEST INFRASTRUCTURE (oracle/_ref recipe): a host stand-in for the handful of CUDA
// language/runtime features the reference's voxlib / gridencoder sources use, so that
// those sources compile UNCHANGED (apart from the <<<>>> launch token, rewritten on the
// fly by oracle/build_ref.py) with g++ and run on the CPU.  Each "CUDA thread" is one
// call of the kernel function with thread-local blockIdx/threadIdx; blocks run in
// parallel under OpenMP.  None of the kernels on the path uses shared memory or
// __syncthreads, so sequential threads inside a block are equivalent.
#ifndef SDN_REF_SHIM_CUDA_RUNTIME_H
#define SDN_REF_SHIM_CUDA_RUNTIME_H
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <math.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

#ifndef CUDART_PI_F
#define CUDART_PI_F 3.141592654f
#endif

struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local uint3 blockIdx, threadIdx;
extern thread_local dim3 blockDim, gridDim;

typedef void* cudaStream_t;
typedef int cudaError_t;
static inline cudaError_t cudaGetDevice(int* d) { *d = 0; return 0; }
static inline cudaError_t cudaGetLastError() { return 0; }
#define C10_CUDA_CHECK(x) (void)(x)

// CUDA's global-namespace integer/float min/max
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

// atomics: the shim runs one block per OpenMP thread; scatter-adds from different blocks
// may collide, so these are real atomics.
static inline float atomicAdd(float* a, float v) {
    float old;
    #pragma omp atomic capture
    { old = *a; *a += v; }
    return old;
}
static inline double atomicAdd(double* a, double v) {
    double old;
    #pragma omp atomic capture
    { old = *a; *a += v; }
    return old;
}

namespace sdn_ref_shim {
template <typename F>
struct Bound {
    dim3 g, b; F f;
    template <typename... A>
    void operator()(A&&... a) const {
        const long nblk = (long)g.x * g.y * g.z;
        #pragma omp parallel for schedule(dynamic, 4)
        for (long i = 0; i < nblk; ++i) {
            gridDim = g; blockDim = b;
            blockIdx.x = (unsigned)(i % g.x); blockIdx.y = (unsigned)((i / g.x) % g.y); blockIdx.z = (unsigned)(i / ((long)g.x * g.y));
            for (unsigned tz = 0; tz < b.z; ++tz)
                for (unsigned ty = 0; ty < b.y; ++ty)
                    for (unsigned tx = 0; tx < b.x; ++tx) {
                        threadIdx.x = tx; threadIdx.y = ty; threadIdx.z = tz;
                        f(a...);
                    }
        }
    }
};
struct Launcher {
    dim3 g, b;
    template <typename... X>
    Launcher(dim3 g_, dim3 b_, X...) : g(g_), b(b_) {}
    template <typename F>
    Bound<F> bind(F f) const { return Bound<F>{g, b, f}; }
};
}  // namespace sdn_ref_shim
#endif
