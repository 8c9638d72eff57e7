// Micro-benchmark: what a "unit" of the MLP kernel costs one in-order wave per SIMD, as a function of its MFMA count and of
// the VALU / ds_read_b128 instructions placed in its MFMA gaps -- the cost model behind the activation schedule of field.hip.
//   unit kinds: 6 x v_mfma_f32_32x32x16_f16 (trunk), 4 x the same on 4 accumulators (colour layers, f16 part),
//               2 x v_mfma_scale_f32_32x32x64_f8f6f4 fp6 (colour layers, correction part)
//   fillers per unit: NV independent VALU (v_fma_f32, 8 chains), NL independent ds_read_b128 (waited for one unit later,
//   lgkmcnt(NL), as the kernel's register ring does), spread evenly over the unit's gaps.
// 4 waves per workgroup, one workgroup per CU, every CU busy.  hipcc --offload-arch=gfx950 -O3 tools/unit_cost_ubench.hip -o tools/unit_cost_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8v __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ void ds_read16(half8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

template <int KIND, int NV, int NL, int FK = 0>
__global__ __launch_bounds__(256, 1) void k(float *out, const half8 *in, int iters, long long *cyc) {
    __shared__ __attribute__((aligned(1024))) char lds[65536];
    constexpr int NM = KIND == 0 ? 6 : KIND == 1 ? 4 : 2;
    const int lane = threadIdx.x & 63;
    half8 a[3][4];
    for (int i = 0; i < 4; i++) a[0][i] = a[1][i] = a[2][i] = in[(lane + i) & 127];
    half8 b = in[64 + lane];
    const u32x4v bw = __builtin_bit_cast(u32x4v, b);
    const i32x8v B6 = {(int)bw[0], (int)bw[1], (int)bw[2], (int)bw[3], (int)bw[0], (int)bw[1], 0, 0};
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = (float)lane * 0.001f + i;
    const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, in[0][0] > 0 ? 1.5f : 1.25f)));
    float2v pk[4];
    for (int i = 0; i < 4; i++) pk[i] = float2v{f[2 * i], f[2 * i + 1]};
    f32x16 extra;
    for (int r = 0; r < 16; r++) extra[r] = lane + r;
    extra = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, extra, 0, 0, 0);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds + lane * 16;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 3; u++) {   // three units per trip: the 3-deep register ring of field.hip (reads 2 units ahead)
            half8(&cur)[4] = a[u], (&nxt)[4] = a[(u + 2) % 3];
            if constexpr (NL > 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NL) : "memory");
            int v = 0, l = 0;
#pragma unroll
            for (int m = 0; m < NM; m++) {
                if constexpr (KIND == 2) {
                    // as mfma_mx of field.hip: an fp6 fragment = two 16-byte ring fragments (6 dwords of codes, scale word, pad)
                    const u32x4v w0 = __builtin_bit_cast(u32x4v, cur[(2 * m) % 4]), w1 = __builtin_bit_cast(u32x4v, cur[(2 * m + 1) % 4]);
                    asm volatile("" ::"v"(cur[(2 * m + 1) % 4]));
                    const i32x8v A = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], 0, 0};
                    acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B6, acc[m], 2, 2, 0, (int)w1[2], 0, 127);
                } else {
                    acc[m % (KIND == 0 ? 2 : 4)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[m % 4], b, acc[m % (KIND == 0 ? 2 : 4)], 0, 0, 0);
                }
                const int v_to = NV * (m + 1) / NM, l_to = NL * (m + 1) / NM;
#pragma unroll
                for (; v < v_to; v++) {
                    float &x = f[v % 8], &y = f[(v + 1) % 8];
                    // (fillers are asm volatile: as plain C++ hipcc sinks them all into the last gap of the loop body)
                    if constexpr (FK == 0) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(x) : "s"(sc));
                    if constexpr (FK == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[v % 4]) : "v"(pk[(v + 1) % 4]));
                    if constexpr (FK == 2) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(y));
                    if constexpr (FK == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                    if constexpr (FK == 4) x = extra[v % 16];                                                      // v_accvgpr_read_b32
                    if constexpr (FK == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
                    if constexpr (FK == 6) asm volatile("v_max3_f32 %0, %0, |%1|, |%1|" : "+v"(x) : "v"(y));
                    if constexpr (FK == 7) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, %0 op_sel_hi:[0,0,1]" : "+v"(x) : "v"(y));
                    if constexpr (FK == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[v % 4]) : "v"(pk[(v + 1) % 4]));
                    if constexpr (FK == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                    if constexpr (FK == 10) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(y));
                    if constexpr (FK == 11) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "s"(sc));
                    if constexpr (FK == 12) asm volatile("v_fma_f32 %0, %0, 0.5, %0" : "+v"(x));
                    if constexpr (FK == 13) asm volatile("v_fma_f32 %0, %0, %1, |%0|" : "+v"(x) : "v"(y));
                    if constexpr (FK == 14) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                    if constexpr (FK == 15) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x) : "s"(sc));
                    if constexpr (FK == 16) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x) : "v"(y));
                    if constexpr (FK == 17) asm volatile("v_fma_mix_f32 %0, %0, %1, |%0|" : "+v"(x) : "v"(y));
                    if constexpr (FK == 18) asm volatile("v_fma_mix_f32 %0, %0, %1, |%0|" : "+v"(x) : "s"(sc));
                }
#pragma unroll
                for (; l < l_to; l++) {
                    if (l == 0) ds_read16<0>(nxt[0], base);
                    if (l == 1) ds_read16<1024>(nxt[1], base);
                    if (l == 2) ds_read16<2048>(nxt[2], base);
                    if (l == 3) ds_read16<3072>(nxt[3], base);
                    if (l == 4) ds_read16<4096>(nxt[0], base);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
    for (int i = 0; i < 8; i++) s += f[i];
    for (int i = 0; i < 4; i++) s += pk[i][0] + pk[i][1];
    for (int i = 0; i < 4; i++) s += (float)a[0][i][0] + (float)a[1][i][0] + (float)a[2][i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int NV, int NL, int FK = 0>
void run(float *out, half8 *in, long long *cyc) {
    const int iters = 4000;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<KIND, NV, NL, FK>), dim3(256), dim3(256), 0, 0, out, in, iters, cyc);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const int nm = KIND == 0 ? 6 : KIND == 1 ? 4 : 2;
    static const char *fk[] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_mix_f32", "v_cvt_pk_f16_f32", "v_accvgpr_read", "v_mov_b32", "v_max3_f32",
                               "v_fma_mixlo_f16", "v_pk_add_f32", "v_add_f32", "v_fma v,v,v", "v_fma v,s,v", "v_fma v,0.5,v", "v_fma v,v,|v|", "v_mul_f32",
                               "v_add v,s", "v_fmac_f32", "v_fma_mix v,v,|v|", "v_fma_mix v,s,|v|"};
    printf("%-22s %2d x %-16s  ds_read_b128 %d : %6.1f cycles per unit (matrix pipe alone: %3d)\n",
           KIND == 0 ? "6 x f16 MFMA (2 acc)" : KIND == 1 ? "4 x f16 MFMA (4 acc)" : "2 x fp6 MFMA (2 acc)", NV, fk[FK], NL,
           (double)c / (iters * 3.0), nm * 32);
}


// ---- does switching between v_mfma_f32_32x32x16_f16 and v_mfma_scale_f32_32x32x64_f8f6f4 cost anything?  RF f16 MFMAs then RX fp6
// MFMAs, repeated; 4 rotating accumulators each; nothing else in the loop
template <int RF, int RX>
__global__ __launch_bounds__(256, 1) void ksw(float *out, const half8 *in, int iters, long long *cyc) {
    const int lane = threadIdx.x & 63;
    half8 a = in[lane], b = in[64 + lane];
    const u32x4v bw = __builtin_bit_cast(u32x4v, b);
    const i32x8v B6 = {(int)bw[0], (int)bw[1], (int)bw[2], (int)bw[3], (int)bw[0], (int)bw[1], 0, 0};
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < RF; m++) {
            acc[m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % 4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < RX; m++) {
            acc[m % 4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B6, B6, acc[m % 4], 2, 2, 0, 127, 0, 127);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int RF, int RX>
void run_sw(float *out, half8 *in, long long *cyc) {
    const int iters = 96000 / (RF + RX);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((ksw<RF, RX>), dim3(256), dim3(256), 0, 0, out, in, iters, cyc);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%3d x f16 MFMA then %3d x fp6 MFMA, repeated: %6.1f cycles per MFMA (32 = matrix pipe busy)\n", RF, RX, (double)c / ((double)iters * (RF + RX)));
}

#define COL(K, NL) \
    run<K, 0, NL, 9>(out, in, cyc); run<K, 6, NL, 9>(out, in, cyc); run<K, 12, NL, 9>(out, in, cyc); run<K, 18, NL, 9>(out, in, cyc); \
    run<K, 24, NL, 9>(out, in, cyc); run<K, 30, NL, 9>(out, in, cyc); run<K, 36, NL, 9>(out, in, cyc); run<K, 42, NL, 9>(out, in, cyc); \
    run<K, 48, NL, 9>(out, in, cyc);
#define ROWS(K) COL(K, 0) COL(K, 4) COL(K, 5)

int main() {
    float *out;
    half8 *in;
    long long *cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&in, 128 * 16);
    hipMalloc(&cyc, 8);
    hipMemset(in, 0, 128 * 16);
    if (getenv("UC_SWITCH")) {
        run_sw<64, 0>(out, in, cyc); run_sw<0, 64>(out, in, cyc); run_sw<64, 32>(out, in, cyc); run_sw<16, 8>(out, in, cyc); run_sw<4, 2>(out, in, cyc);
        run_sw<2, 1>(out, in, cyc); run_sw<1, 1>(out, in, cyc); run_sw<8, 8>(out, in, cyc);
        return 0;
    }
    ROWS(0) ROWS(1) ROWS(2)
    run<0, 24, 0, 1>(out, in, cyc); run<0, 24, 0, 2>(out, in, cyc); run<0, 24, 0, 3>(out, in, cyc); run<0, 24, 0, 4>(out, in, cyc);
    run<0, 24, 0, 5>(out, in, cyc); run<0, 24, 0, 6>(out, in, cyc); run<0, 24, 0, 7>(out, in, cyc); run<0, 24, 0, 8>(out, in, cyc);
    run<0, 24, 0, 10>(out, in, cyc); run<0, 24, 0, 11>(out, in, cyc); run<0, 24, 0, 12>(out, in, cyc); run<0, 24, 0, 13>(out, in, cyc);
    run<0, 24, 0, 14>(out, in, cyc); run<0, 24, 0, 15>(out, in, cyc); run<0, 24, 0, 16>(out, in, cyc); run<0, 24, 0, 17>(out, in, cyc); run<0, 24, 0, 18>(out, in, cyc);
    run<0, 24, 0, 9>(out, in, cyc); run<0, 48, 0, 9>(out, in, cyc); run<0, 48, 0, 5>(out, in, cyc);
    return 0;
}
