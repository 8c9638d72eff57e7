// Micro-benchmark 2: one vs two waves per SIMD, v_mfma_f32_16x16x32_f16 and 32x32x16, with VALU / LDS fillers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NFILL_V, int NFILL_L>
__global__ void k(float *out, const half8 *in, int iters) {
    __shared__ char lds[65536];
    const int lane = threadIdx.x & 63;
    half8 a = in[lane], b = in[64 + lane];
    f32x16 acc32[2];
    f32x4 acc16[4];
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) acc32[i][r] = 0.f;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 4; r++) acc16[i][r] = 0.f;
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = (float)lane * 0.001f + i;
    half8 d[2] = {a, b};
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (SHAPE == 32) acc32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, d[m & 1], acc32[m & 1], 0, 0, 0);
            else acc16[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, d[m & 1], acc16[m & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NFILL_V; q++) f[q % 8] = f[q % 8] * 1.0001f + 0.5f;
#pragma unroll
            for (int q = 0; q < NFILL_L; q++)
                d[q % 2] = *reinterpret_cast<const half8 *>(lds + ((lane * 16 + (m * 2 + q) * 1024) & 65535));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) s += acc32[i][r];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 4; r++) s += acc16[i][r];
    for (int i = 0; i < 8; i++) s += f[i];
    s += (float)d[0][0] + (float)d[1][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NV, int NL>
void run(int threads, float *out, half8 *in) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<SHAPE, NV, NL>), dim3(256), dim3(threads), 0, 0, out, in, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NV, NL>), dim3(256), dim3(threads), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (SHAPE == 32 ? 32768.0 : 16384.0);
    const double n_mfma = (double)iters * 8 * (threads / 64) * 256;
    printf("mfma %dx%d  waves/SIMD=%d  valu/mfma=%d lds/mfma=%d : %6.2f ns per MFMA per SIMD-slot, %7.1f TFLOP/s\n", SHAPE, SHAPE,
           threads / 256, NV, NL, ms * 1e6 / (iters * 8.0 * (threads / 256)), n_mfma * flop / (ms * 1e-3) / 1e12);
}

#define GRID(SH) \
    run<SH, 0, 0>(256, out, in); run<SH, 4, 0>(256, out, in); run<SH, 4, 1>(256, out, in); run<SH, 8, 1>(256, out, in); \
    run<SH, 0, 0>(512, out, in); run<SH, 4, 0>(512, out, in); run<SH, 4, 1>(512, out, in); run<SH, 8, 1>(512, out, in); run<SH, 12, 1>(512, out, in); \
    run<SH, 0, 0>(768, out, in); run<SH, 8, 1>(768, out, in);

int main() {
    float *out; half8 *in;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 128 * 16);
    hipMemset(in, 0, 128 * 16);
    GRID(32)
    GRID(16)
    return 0;
}
