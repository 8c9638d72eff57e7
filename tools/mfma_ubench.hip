// Micro-benchmark: how many filler instructions of which kind hide behind v_mfma_f32_32x32x16_f16 when ONE wave
// runs per SIMD, for 2 vs 4 rotating accumulators.   hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o tools/mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int KIND, int NFILL>
__global__ __launch_bounds__(256, 1) void k(float *out, const half8 *in, int iters, long long *cyc) {
    __shared__ char lds[65536];
    const int lane = threadIdx.x & 63;
    half8 a = in[lane], b = in[64 + lane];
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = (float)lane * 0.001f + i;
    half8 d[2] = {a, b};
    f32x16 extra;                       // a separate accumulator block to read with v_accvgpr_read
    for (int r = 0; r < 16; r++) extra[r] = lane + r;
    extra = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, extra, 0, 0, 0);
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NFILL; q++) {
                if (KIND == 0) f[q % 8] = f[q % 8] * 1.0001f + 0.5f;                       // independent v_fma chains (8 chains)
                if (KIND == 1) f[q % 8] += extra[(m * NFILL + q) % 16];                    // v_accvgpr_read + add
                if (KIND == 2) d[q % 2] = *reinterpret_cast<const half8 *>(lds + ((lane * 16 + (m * NFILL + q) * 1024) & 65535));  // ds_read_b128
                if (KIND == 3) { auto p = __builtin_amdgcn_cvt_pkrtz(f[q % 8], f[(q + 1) % 8]); f[q % 8] = (float)p[0] + f[q % 8]; }  // cvt_pkrtz + cvt_f32_f16 + add (dependent triple)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    for (int i = 0; i < 8; i++) s += f[i];
    s += (float)d[0][0] + (float)d[1][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int KIND, int NFILL>
void run(const char *name, float *out, half8 *in, long long *cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, KIND, NFILL>), dim3(256), dim3(256), 0, 0, out, in, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, KIND, NFILL>), dim3(256), dim3(256), 0, 0, out, in, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-14s nacc=%d fill/mfma=%d : %6.1f ns/mfma (wall)  %6.1f memtime-ticks/mfma\n", name, NACC, NFILL,
           ms * 1e6 / (iters * 8.0), (double)c / (iters * 8.0));
}

#define ROW(K, NAME) \
    run<2, K, 0>(NAME, out, in, cyc); run<2, K, 2>(NAME, out, in, cyc); run<2, K, 4>(NAME, out, in, cyc); run<2, K, 6>(NAME, out, in, cyc); run<2, K, 8>(NAME, out, in, cyc); \
    run<4, K, 0>(NAME, out, in, cyc); run<4, K, 4>(NAME, out, in, cyc); run<4, K, 8>(NAME, out, in, cyc);

int main() {
    float *out; half8 *in; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&in, 128 * 16); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 128 * 16);
    ROW(0, "v_fma")
    ROW(1, "accvgpr_read")
    ROW(2, "ds_read_b128")
    ROW(3, "cvt chain")
    run<1, 0, 0>("v_fma", out, in, cyc);
    return 0;
}
