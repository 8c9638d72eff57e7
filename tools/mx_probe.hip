// Probe of the gfx950 MX (block-scaled) fp6 machinery the field MLP's colour layers rely on: element order and scale
// semantics of v_cvt_scalef32_2xpk16_fp6_f32 / v_cvt_scalef32_pk32_fp6_f16, operand layout and scale operands of
// v_mfma_scale_f32_32x32x64_f8f6f4 (cbsz = blgp = 2: fp6 e2m3).  Stand-alone: hipcc --offload-arch=gfx950 -o mx_probe mx_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));

static float dec6(int c) {   // fp6 e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m * 0.125f : (1.f + m * 0.125f) * (float)(1 << (e - 1));
    return s ? -v : v;
}

__global__ void cvt_kernel(const float *in, unsigned *out, float scale) {
    f32x16 a, b;
    for (int i = 0; i < 16; i++) { a[i] = in[i]; b[i] = in[16 + i]; }
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int i = 0; i < 6; i++) out[i] = r[i];
    h32 hv;
    for (int i = 0; i < 32; i++) hv[i] = (_Float16)in[i];
    const u32x6 r2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, scale);
    for (int i = 0; i < 6; i++) out[6 + i] = r2[i];
}

template <int OA, int OB>
__global__ void mfma_kernel(const int *A, const int *B, float *C, const int *sa, const int *sb) {
    i32x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = A[threadIdx.x * 8 + i]; b[i] = B[threadIdx.x * 8 + i]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, OA, sa[threadIdx.x], OB, sb[threadIdx.x]);
    for (int i = 0; i < 16; i++) C[threadIdx.x * 16 + i] = c[i];
}

static void decode(const unsigned *w, int *codes) {
    for (int p = 0; p < 32; p++) {
        const int bit = 6 * p, d = bit / 32, o = bit % 32;
        unsigned long long two = w[d] | ((unsigned long long)(d + 1 < 6 ? w[d + 1] : 0) << 32);
        codes[p] = (int)((two >> o) & 63);
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    // ---- 1. conversions -------------------------------------------------------------------------------------------
    float h_in[32];
    for (int i = 0; i < 32; i++) h_in[i] = dec6(i);           // code i <-> value: position p of the result holds code of element ...
    float *d_in; unsigned *d_out;
    CK(hipMalloc(&d_in, sizeof(h_in))); CK(hipMalloc(&d_out, 12 * 4));
    const float scales[] = {1.f, 2.f, 0.5f, 3.f};
    for (float sc : scales) {
        float in2[32];
        for (int i = 0; i < 32; i++) in2[i] = h_in[i] * (sc == 3.f ? 2.f : sc) * (i == 5 ? -1.f : 1.f);   // element 5 negative
        CK(hipMemcpy(d_in, in2, sizeof(in2), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(1), 0, 0, d_in, d_out, sc);
        unsigned h_out[12];
        CK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
        int c[32];
        decode(h_out, c);
        printf("cvt 2xpk16_fp6_f32 scale %g (inputs = dec6(i) * %g): position -> code:", sc, sc == 3.f ? 2.f : sc);
        for (int p = 0; p < 32; p++) printf(" %d", c[p]);
        printf("\n");
        decode(h_out + 6, c);
        printf("cvt pk32_fp6_f16   scale %g                        : position -> code:", sc);
        for (int p = 0; p < 32; p++) printf(" %d", c[p]);
        printf("\n");
    }
    // rounding / saturation: a few awkward values at scale 1
    {
        const float t[32] = {7.5f, 7.74f, 7.76f, 8.f, 100.f, 0.0624f, 0.0626f, 0.1875f, 0.19f, 3.874f, 3.876f, 1.0624f, 1.0626f, -7.9f, 1e-9f, 6.25f,
                             6.f, 6.75f, 7.25f, 0.3125f, 0.4375f, 2.125f, 2.375f, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        CK(hipMemcpy(d_in, t, sizeof(t), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(1), 0, 0, d_in, d_out, 1.f);
        unsigned h_out[12];
        CK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
        int c[32];
        decode(h_out, c);
        printf("rounding (element i, interleaving as found above): ");
        for (int p = 0; p < 32; p++) printf(" [%d]=%g", p, dec6(c[p]));
        printf("\n");
    }
    // small inputs with a small scale: the lo block of the field MLP (|x - f16(x)| <= 2^-11 |x|, scale 2^(e - 13))
    for (int k = 9; k <= 15; k += 2) {
        float t[32];
        const float sc = std::ldexp(1.f, -k);
        for (int i = 0; i < 32; i++) t[i] = dec6(i) * sc * ((i & 1) ? -1.f : 1.f);
        CK(hipMemcpy(d_in, t, sizeof(t), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(1), 0, 0, d_in, d_out, sc);
        unsigned h_out[12];
        CK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
        int c[32], bad16 = 0, bad32 = 0;
        decode(h_out + 6, c);
        for (int p = 0; p < 32; p++) bad16 += c[p] != (p | ((p & 1) && p ? 32 : 0));
        decode(h_out, c);
        for (int p = 0; p < 32; p++) { const int i = (p & 1) ? 16 + p / 2 : p / 2; bad32 += c[p] != (i | ((i & 1) && i ? 32 : 0)); }
        printf("scale 2^-%d, inputs dec6(i) * 2^-%d (f16 inputs become subnormal below 2^-14): wrong codes pk32_fp6_f16 %d, 2xpk16_fp6_f32 %d\n", k, k, bad16, bad32);
    }
    // ---- 2. MFMA ----------------------------------------------------------------------------------------------------
    std::vector<int> Ac(32 * 64), Bc(64 * 32);
    srand(1);
    for (auto &v : Ac) v = rand() & 63;
    for (auto &v : Bc) v = rand() & 63;
    auto pack = [](const std::vector<int> &codes, bool isA, std::vector<int> &out) {
        out.assign(64 * 8, 0);
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 32; i++) {
                const int k = 32 * (l >> 5) + i, rc = l & 31;
                const unsigned code = isA ? codes[rc * 64 + k] : codes[k * 32 + rc];
                const int bit = 6 * i, d = bit / 32, o = bit % 32;
                unsigned long long two = (unsigned long long)code << o;
                out[l * 8 + d] |= (int)(unsigned)(two & 0xffffffffu);
                if (o > 26) out[l * 8 + d + 1] |= (int)(unsigned)(two >> 32);
            }
    };
    std::vector<int> Ap, Bp;
    pack(Ac, true, Ap);
    pack(Bc, false, Bp);
    int *dA, *dB, *dsa, *dsb; float *dC;
    CK(hipMalloc(&dA, 64 * 8 * 4)); CK(hipMalloc(&dB, 64 * 8 * 4)); CK(hipMalloc(&dC, 64 * 16 * 4)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
    CK(hipMemcpy(dA, Ap.data(), 64 * 8 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, Bp.data(), 64 * 8 * 4, hipMemcpyHostToDevice));
    for (int test = 0; test < 6; test++) {
        int sa[64], sb[64], ea[64], eb[64];
        for (int l = 0; l < 64; l++) {
            ea[l] = test == 0 ? 0 : (l % 3) - 1;            // exponent of the lane's A block (row l&31, k-half l>>5)
            eb[l] = test == 0 ? 0 : (l >> 5) + ((l & 31) % 2);
            const int ba = 127 + ea[l], bb = 127 + eb[l];
            // tests 0/1: scale in byte 0; test 2: scale in byte 1 (opsel 1), byte 0 = junk; test 3: byte 2 / byte 3
            // test 4: A byte 0, B byte 1 (byte 0 = junk); test 5: A byte 1, B byte 0
            sa[l] = test <= 1 ? ba : (test == 2 ? (ba << 8) | 0x11 : test == 3 ? (ba << 16) | 0x2211 : test == 4 ? ba | 0x6600 : (ba << 8) | 0x11);
            sb[l] = test <= 1 ? bb : (test == 2 ? (bb << 8) | 0x33 : test == 3 ? (bb << 24) | 0x554433 : test == 4 ? (bb << 8) | 0x33 : bb | 0x7700);
        }
        CK(hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice));
        if (test <= 1) hipLaunchKernelGGL((mfma_kernel<0, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
        else if (test == 2) hipLaunchKernelGGL((mfma_kernel<1, 1>), dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
        else if (test == 3) hipLaunchKernelGGL((mfma_kernel<2, 3>), dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
        else if (test == 4) hipLaunchKernelGGL((mfma_kernel<0, 1>), dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
        else hipLaunchKernelGGL((mfma_kernel<1, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
        float hC[64 * 16];
        CK(hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
                double ref = 0;
                for (int kb = 0; kb < 2; kb++) {
                    double s = 0;
                    for (int i = 0; i < 32; i++) s += (double)dec6(Ac[row * 64 + 32 * kb + i]) * dec6(Bc[(32 * kb + i) * 32 + col]);
                    ref += s * std::ldexp(1.0, ea[row + 32 * kb] + eb[col + 32 * kb]);
                }
                maxerr = std::fmax(maxerr, std::fabs(ref - hC[l * 16 + r]));
                maxref = std::fmax(maxref, std::fabs(ref));
            }
        printf("mfma_scale fp6 test %d: max |D - ref| = %g (max |ref| %g) %s\n", test, maxerr, maxref, maxerr < 1e-3 * maxref ? "OK" : "MISMATCH");
    }
    return 0;
}
