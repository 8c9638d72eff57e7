// Micro-benchmark 4: LDS-DMA (global_load_lds_dwordx4) throughput per CU when the source streams from L2 (a 1.5 MB buffer
// shared by all workgroups, like the packed MLP weights), with and without MFMAs in between; vs plain global_load_dwordx4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const char glb_char;
constexpr int WBYTES = 92 * 16384;

// MODE 0: DMA; MODE 1: plain loads into registers (summed); NM = MFMAs per 16 KiB slot per wave
template <int MODE, int NM>
__global__ __launch_bounds__(256, 1) void k(float *out, const half8 *in, const char *wsrc, int iters) {
    __shared__ __attribute__((aligned(1024))) char lds[131072];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 a = in[lane], b = in[64 + lane];
    f32x16 acc[2];
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    int slot = 0;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        const char *src = wsrc + (size_t)slot * 16384 + wave * 4096 + lane * 16;
        char *dst = lds + (it & 7) * 16384 + wave * 4096;
        if (MODE == 0) {
            __builtin_amdgcn_global_load_lds((glb_char *)src, (lds_char *)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_char *)src, (lds_char *)dst, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds((glb_char *)src, (lds_char *)dst, 16, 2048, 0);
            __builtin_amdgcn_global_load_lds((glb_char *)src, (lds_char *)dst, 16, 3072, 0);
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        } else {
            f32x4 v0, v1, v2, v3;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
                         "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(src) : "memory");
            sum += v0 + v1 + v2 + v3;
        }
        slot = slot + 1 == 92 ? 0 : slot + 1;
#pragma unroll
        for (int m = 0; m < NM; m++) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 1], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = sum[0] + sum[1] + sum[2] + sum[3];
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NM>
void run(float *out, half8 *in, char *w) {
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE, NM>), dim3(256), dim3(256), 0, 0, out, in, w, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NM>), dim3(256), dim3(256), 0, 0, out, in, w, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("ubench4 %s mfma/slot=%2d : %7.1f ns per 16 KiB slot  -> %6.1f GB/s per CU, %5.2f TB/s total\n", MODE ? "plain-load" : "lds-dma   ", NM,
           ms * 1e6 / iters, 16384.0 / (ms * 1e6 / iters), 16384.0 * 256 / (ms * 1e6 / iters) / 1000);
}

int main() {
    float *out; half8 *in; char *w;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 128 * 16); hipMalloc(&w, WBYTES);
    hipMemset(in, 0, 128 * 16); hipMemset(w, 0, WBYTES);
    run<0, 0>(out, in, w); run<0, 8>(out, in, w); run<0, 16>(out, in, w); run<0, 24>(out, in, w); run<0, 32>(out, in, w);
    run<1, 0>(out, in, w); run<1, 24>(out, in, w);
    return 0;
}
