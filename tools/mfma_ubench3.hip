// Micro-benchmark 3: issue cost of INDEPENDENT ds_read_b128 / global_load_lds between MFMAs (results never waited for
// inside the loop), one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const char glb_char;

template <int NL, int ND, int NV>
__global__ void k(float *out, const half8 *in, const char *wsrc, int iters) {
    __shared__ __attribute__((aligned(1024))) char lds[131072];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 a = in[lane], b = in[64 + lane];
    f32x16 acc[2];
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    half8 d[8];
    float f[8];
    for (int i = 0; i < 8; i++) { d[i] = a; f[i] = lane * 0.01f + i; }
    const unsigned base = (unsigned)(size_t)(lds_char *)lds + lane * 16;
    const char *src = wsrc + (size_t)blockIdx.x * 65536 + wave * 4096 + lane * 16;
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 1], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NL; q++)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[(m * NL + q) & 7]) : "v"(base), "n"(((m * 2 + q) & 15) * 1024));
            if (ND > 0 && (m % (8 / ND)) == 0) {
                __builtin_amdgcn_global_load_lds((glb_char *)src, (lds_char *)(lds + 65536 + wave * 8192 + (m & 7) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NV; q++) f[q % 8] = f[q % 8] * 1.0001f + 0.5f;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ND > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    for (int i = 0; i < 8; i++) s += (float)d[i][0] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NL, int ND, int NV>
void run(int threads, float *out, half8 *in, char *w) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<NL, ND, NV>), dim3(256), dim3(threads), 0, 0, out, in, w, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NL, ND, NV>), dim3(256), dim3(threads), 0, 0, out, in, w, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("ubench3 waves/SIMD=%d  ds_read/mfma=%d  dma/8mfma=%d valu/mfma=%d : %6.2f ns per MFMA per wave-slot\n", threads / 256, NL, ND, NV,
           ms * 1e6 / (iters * 8.0 * (threads / 256)));
}

int main() {
    float *out; half8 *in; char *w;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 128 * 16); hipMalloc(&w, 256 * 65536);
    hipMemset(in, 0, 128 * 16); hipMemset(w, 0, 256 * 65536);
    run<0, 0, 0>(256, out, in, w); run<1, 0, 0>(256, out, in, w); run<2, 0, 0>(256, out, in, w); run<1, 0, 2>(256, out, in, w);
    run<0, 1, 0>(256, out, in, w); run<0, 2, 0>(256, out, in, w); run<0, 4, 0>(256, out, in, w); run<1, 2, 2>(256, out, in, w);
    run<0, 0, 0>(512, out, in, w); run<1, 0, 0>(512, out, in, w); run<2, 0, 0>(512, out, in, w); run<1, 0, 2>(512, out, in, w);
    run<0, 2, 0>(512, out, in, w); run<0, 4, 0>(512, out, in, w); run<1, 2, 2>(512, out, in, w); run<1, 4, 0>(512, out, in, w);
    return 0;
}
