// Multi-resolution hash-grid feature encoder (forward, dy_dx, backward) for gfx950.
//
// Contract: _gridencoder.grid_encode_forward / grid_encode_backward of the
// reference (gridencoder/src/gridencoder.cu: fast_hash :35-51, get_grid_index
// :54-72, kernel_grid :75-224, kernel_grid_backward :227-314,
// kernel_input_backward :317-343, entry points :423-478).
//
// MI355X mapping of the forward:
//   * the op is a pure gather: per (sample, level) 2^D corner rows of C values
//     are fetched from a <=16 MB per-level table and blended.  It is bound by
//     L2 / Infinity-Cache / HBM bandwidth, never by ALU, so the kernel is
//     organised around memory:
//   * one lane owns one (sample, level); a corner row is fetched with the
//     widest load the row allows (C=8 f32 -> 2 x dwordx4, C=8 f16 -> 1 x dwordx4)
//     and all 2^D row loads of a lane are independent so dozens are in flight
//     per lane before the first blend;
//   * levels are the slow grid dimension and sample blocks are remapped so
//     that each of the 8 XCDs (private 4 MiB L2 each) walks one contiguous
//     eighth of the samples: at any time an XCD's L2 only has to hold the part
//     of ONE level's table that its (spatially coherent) samples touch;
//   * outputs [L,B,C] are written as one contiguous C-row per lane, i.e. fully
//     coalesced 32 B x 64 lanes; indices are 64-bit (a 4K frame has 339 M
//     samples; the reference's uint32 offsets overflow there, gridencoder.cu:87-96).
#include <hip/hip_fp16.h>

#include <type_traits>

#include <cstdlib>

#include "sdn_common.h"

namespace {

constexpr int NXCD = 8;
constexpr int FWD_THREADS = 256;
constexpr uint32_t MAX_TABLE_LEVELS = 32;

// Per-level scale / resolution (gridencoder.cu:126-127) evaluated ONCE on the
// host with libm's exp2f and handed to the kernels by value, so that the
// drop-in op, the fused renderer and the CPU oracle all see the same constants
// (the device's v_exp_f32 is a 1-ulp approximation).  More than 32 levels fall
// back to the in-kernel formula.
struct GridLevels {
    float scale[MAX_TABLE_LEVELS];
    uint32_t resolution[MAX_TABLE_LEVELS];
    uint32_t use_table;
};

inline GridLevels make_levels(uint32_t L, float S, uint32_t H) {
    GridLevels lv;
    lv.use_table = L <= MAX_TABLE_LEVELS;
    for (uint32_t l = 0; l < MAX_TABLE_LEVELS; l++) {
        const float sc = exp2f((float)l * S) * (float)H - 1.0f;
        lv.scale[l] = sc;
        lv.resolution[l] = (uint32_t)ceilf(sc) + 1;
    }
    return lv;
}

__device__ __forceinline__ void level_params(const GridLevels &lv, uint32_t level, float S, uint32_t H, float &scale,
                                             uint32_t &resolution) {
    if (lv.use_table) {
        scale = lv.scale[level];
        resolution = lv.resolution[level];
    } else {
        scale = exp2f(level * S) * H - 1.0f;
        resolution = (uint32_t)ceilf(scale) + 1;
    }
}

// a * b + c with two roundings (no FMA contraction), whatever the translation unit's -ffp-contract:
// HIP's __fmul_rn/__fadd_rn are plain `*` / `+` and still get fused.
__device__ __forceinline__ float mul_add_exact(float a, float b, float c) {
#pragma clang fp contract(off)
    const float p = a * b;
    return p + c;
}

template <uint32_t D>
__device__ __forceinline__ uint32_t fast_hash(const uint32_t (&pg)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t h = 0;
#pragma unroll
    for (uint32_t i = 0; i < D; ++i) h ^= pg[i] * primes[i];
    return h;
}

// Row index (not yet multiplied by C) of a grid vertex.  `dense` is decided per
// level on the host side of the kernel (uniform), mirroring the reference's
// stride walk: dims are linearised while stride <= hashmap_size, and the hash
// replaces the partial sum only when gridtype==hash and the walk overflowed.
template <uint32_t D>
__device__ __forceinline__ uint32_t grid_row(const uint32_t (&pg)[D], uint32_t gridtype, uint32_t dim_stride,
                                             uint32_t hashmap_size) {
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pg[d] * stride;
            stride *= dim_stride;
        }
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash<D>(pg);
    return index % hashmap_size;
}

template <uint32_t C>
__device__ __forceinline__ void load_row(const float *__restrict__ g, float (&v)[C]) {
    if constexpr (C == 8) {
        const float4 a = *reinterpret_cast<const float4 *>(g);
        const float4 b = *reinterpret_cast<const float4 *>(g + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if constexpr (C == 4) {
        const float4 a = *reinterpret_cast<const float4 *>(g);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    } else if constexpr (C == 2) {
        const float2 a = *reinterpret_cast<const float2 *>(g);
        v[0] = a.x; v[1] = a.y;
    } else {
        v[0] = g[0];
    }
}

template <uint32_t C>
__device__ __forceinline__ void load_row(const __half *__restrict__ g, float (&v)[C]) {
    if constexpr (C == 8) {
        const uint4 a = *reinterpret_cast<const uint4 *>(g);
        const __half2 *h = reinterpret_cast<const __half2 *>(&a);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float2 f = __half22float2(h[i]);
            v[2 * i] = f.x;
            v[2 * i + 1] = f.y;
        }
    } else if constexpr (C == 4) {
        const uint2 a = *reinterpret_cast<const uint2 *>(g);
        const __half2 *h = reinterpret_cast<const __half2 *>(&a);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float2 f = __half22float2(h[i]);
            v[2 * i] = f.x;
            v[2 * i + 1] = f.y;
        }
    } else if constexpr (C == 2) {
        const float2 f = __half22float2(*reinterpret_cast<const __half2 *>(g));
        v[0] = f.x; v[1] = f.y;
    } else {
        v[0] = __half2float(g[0]);
    }
}

template <uint32_t C>
__device__ __forceinline__ void store_row(float *__restrict__ o, const float (&v)[C]) {
    if constexpr (C == 8) {
        *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else if constexpr (C == 4) {
        *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (C == 2) {
        *reinterpret_cast<float2 *>(o) = make_float2(v[0], v[1]);
    } else {
        o[0] = v[0];
    }
}

template <uint32_t C>
__device__ __forceinline__ void store_row(__half *__restrict__ o, const float (&v)[C]) {
#pragma unroll
    for (uint32_t i = 0; i < C; i++) o[i] = __float2half(v[i]);
}

// XCD-aware sample-block remap: hardware places workgroup w on XCD w % 8; give
// every XCD a contiguous range of sample blocks (bijective for any n).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t w, uint32_t n) {
    const uint32_t q = n / NXCD, r = n % NXCD;
    const uint32_t xcd = w % NXCD, idx = w / NXCD;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// float -> nearest half value, as a float (the empty asm keeps hipcc from fusing the round trip into a wider expression)
__device__ __forceinline__ float round_h(float x) {
    float r = __half2float(__float2half(x));
    asm("" : "+v"(r));
    return r;
}

// `results[ch] += w * value` in the reference's scalar_t arithmetic (gridencoder.cu:143, :170; :184, :215).
// float: as written.  at::Half: `results` is a half -- the float product is rounded to half by the implicit Half(float),
// then the half + half sum rounds again (c10::Half operator+); bit-identical to the reference's own kernel compiled for
// the host (tests/test_ref_pin_cpu.py::test_grid_forward_half_oracle_equals_reference_source).
template <typename T>
__device__ __forceinline__ void acc_scalar_t(float &res, float w, float v) {
    if constexpr (sizeof(T) == 2) {
        float p = w * v;            // a float product first (rounded to f32): fused into the conversion (v_fma_mixlo_f16) it would
        asm("" : "+v"(p));          // round ONCE to half and differ from the reference's two roundings by an ulp now and then
        res = round_h(res + round_h(p));
    } else {
        res += w * v;
    }
}

template <typename T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(FWD_THREADS) void grid_fwd_kernel(const float *__restrict__ inputs,
                                                              const T *__restrict__ grid,
                                                              const int32_t *__restrict__ offsets,
                                                              T *__restrict__ outputs, uint32_t B, uint32_t L, float S,
                                                              uint32_t H, bool calc_grad_inputs, T *__restrict__ dy_dx,
                                                              uint32_t gridtype, bool align_corners,
                                                              const GridLevels lv) {
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint64_t b = (uint64_t)blk * FWD_THREADS + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;

    grid += (uint64_t)(uint32_t)offsets[level] * C;
    const float *in = inputs + b * D;
    T *out = outputs + ((uint64_t)level * B + b) * C;

    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = in[d];
        oob |= (x[d] < 0 || x[d] > 1);  // gridencoder.cu:99-106
    }
    if (oob) {
        float z[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) z[c] = 0.f;
        store_row<C>(out, z);
        if (calc_grad_inputs) {
            T *dd = dy_dx + (b * L + level) * (uint64_t)(D * C);
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) dd[i] = (T)0.f;
        }
        return;
    }

    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    float scale;
    uint32_t resolution;
    level_params(lv, level, S, H, scale, resolution);  // :126-127
    const uint32_t dim_stride = align_corners ? resolution : resolution + 1;

    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {  // :133-138
        // un-contracted multiply-add: the cell index and the fractional position are then
        // bit-identical to the oracle (an FMA here moves pos by up to 1 ulp = 1.2e-4 at scale 2047,
        // which shows up as ~6e-5 in the blended feature)
        pos[d] = mul_add_exact(x[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= (float)pg[d];
    }

    float res[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) res[c] = 0.f;

#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {  // :146-171, same corner and multiply order
        float w = 1.f;
        uint32_t pgl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) {
                w *= 1 - pos[d];
                pgl[d] = pg[d];
            } else {
                w *= pos[d];
                pgl[d] = pg[d] + 1;
            }
        }
        const uint32_t row = grid_row<D>(pgl, gridtype, dim_stride, hashmap_size);
        float v[C];
        load_row<C>(grid + (uint64_t)row * C, v);
#pragma unroll
        for (uint32_t c = 0; c < C; c++) acc_scalar_t<T>(res[c], w, v[c]);
    }
    store_row<C>(out, res);

    if (calc_grad_inputs) {  // :181-223
        T *dd = dy_dx + (b * L + level) * (uint64_t)(D * C);
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float rg[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) rg[c] = 0.f;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float w = scale;
                uint32_t pgl[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) {
                        w *= 1 - pos[d];
                        pgl[d] = pg[d];
                    } else {
                        w *= pos[d];
                        pgl[d] = pg[d] + 1;
                    }
                }
                pgl[gd] = pg[gd];
                const uint32_t rl = grid_row<D>(pgl, gridtype, dim_stride, hashmap_size);
                pgl[gd] = pg[gd] + 1;
                const uint32_t rr = grid_row<D>(pgl, gridtype, dim_stride, hashmap_size);
                float vl[C], vr[C];
                load_row<C>(grid + (uint64_t)rl * C, vl);
                load_row<C>(grid + (uint64_t)rr * C, vr);
#pragma unroll
                for (uint32_t c = 0; c < C; c++)   // at::Half: grid[right] - grid[left] is a half subtraction
                    acc_scalar_t<T>(rg[c], w, sizeof(T) == 2 ? round_h(vr[c] - vl[c]) : vr[c] - vl[c]);   // (half - half is exact in f32)
            }
#pragma unroll
            for (uint32_t c = 0; c < C; c++) dd[gd * C + c] = (T)rg[c];
        }
    }
}

// Forward without dy_dx, f32 tables (inference: what GridEncoder.forward runs in the un-fused path), wave-cooperative:
// FOUR lanes share one (sample, level).  Lane q of the quad takes the corners whose two lowest dimension bits are q --
// 2^(D-2) rows of C floats in flight per lane instead of 2^D behind one another, 4x the waves to hide the L2 / Infinity
// Cache latency of the gathers -- blends them with the reference's weights (same factor order per corner), and the four
// partial rows are added with two DPP quad permutes.  Each lane then stores its C/4 channels: one coalesced row per quad.
// Only the summation order differs from the reference (grouped by quad lane instead of corner order: <= a few ulp).
template <int CTRL>
__device__ __forceinline__ float quad_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, o);
}

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(FWD_THREADS) void grid_fwd_quad_kernel(const float *__restrict__ inputs,
                                                                   const float *__restrict__ grid,
                                                                   const int32_t *__restrict__ offsets,
                                                                   float *__restrict__ outputs, uint32_t B, uint32_t L, float S,
                                                                   uint32_t H, uint32_t gridtype, bool align_corners,
                                                                   const GridLevels lv) {
    static_assert(D >= 2, "two dimensions are split over the quad");
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t q = threadIdx.x & 3;
    const uint64_t b0 = (uint64_t)blk * (FWD_THREADS / 4) + (threadIdx.x >> 2);
    const bool live = b0 < B;                      // whole quads are live or not; dead lanes follow along for the DPP adds
    const uint64_t b = live ? b0 : (uint64_t)B - 1;
    const uint32_t level = blockIdx.y;
    grid += (uint64_t)(uint32_t)offsets[level] * C;
    const float *in = inputs + b * D;

    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = in[d];
        oob |= (x[d] < 0 || x[d] > 1);  // gridencoder.cu:99-106
    }
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    float scale;
    uint32_t resolution;
    level_params(lv, level, S, H, scale, resolution);
    const uint32_t dim_stride = align_corners ? resolution : resolution + 1;
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = mul_add_exact(oob ? 0.f : x[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= (float)pg[d];
    }
    float res[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) res[c] = 0.f;
#pragma unroll
    for (uint32_t hi = 0; hi < (1u << (D - 2)); hi++) {
        const uint32_t idx = (hi << 2) | q;
        float w = 1.f;
        uint32_t pgl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {   // the reference's factor order, :152-160
            if ((idx & (1u << d)) == 0) {
                w *= 1 - pos[d];
                pgl[d] = pg[d];
            } else {
                w *= pos[d];
                pgl[d] = pg[d] + 1;
            }
        }
        const uint32_t row = grid_row<D>(pgl, gridtype, dim_stride, hashmap_size);
        float v[C];
        load_row<C>(grid + (uint64_t)row * C, v);
#pragma unroll
        for (uint32_t c = 0; c < C; c++) res[c] += w * v[c];
    }
#pragma unroll
    for (uint32_t c = 0; c < C; c++) {
        res[c] = quad_add<0xB1>(res[c]);   // quad_perm [1,0,3,2]
        res[c] = quad_add<0x4E>(res[c]);   // quad_perm [2,3,0,1]
        if (oob) res[c] = 0.f;
    }
    if (!live) return;
    float *out = outputs + ((uint64_t)level * B + b) * C;
    if constexpr (C == 8) {
        // lane q stores channels 2q, 2q+1 (a constant-index select chain: no dynamic register indexing)
        const float lo = q == 0 ? res[0] : q == 1 ? res[2] : q == 2 ? res[4] : res[6];
        const float hi = q == 0 ? res[1] : q == 1 ? res[3] : q == 2 ? res[5] : res[7];
        *reinterpret_cast<float2 *>(out + 2 * q) = make_float2(lo, hi);
    } else if constexpr (C == 4) {
        out[q] = q == 0 ? res[0] : q == 1 ? res[1] : q == 2 ? res[2] : res[3];
    } else if constexpr (C == 2) {
        if (q < 2) out[q] = q == 0 ? res[0] : res[1];
    } else {
        if (q == 0) out[0] = res[0];
    }
}

// Scatter of output gradients into the table: one lane per (sample, level),
// all C channels (the reference uses N_C = 2 channels per thread; with f32
// hardware atomics a full 32 B row per lane keeps the L2 atomic units fed with
// adjacent addresses).  Order of accumulation is nondeterministic, as in the
// reference.
// f16 (gridencoder.cu:296-304): the two contributions of a channel pair are rounded to half ((__half)(w * g)) and
// added with ONE packed atomic (global_atomic_pk_add_f16), i.e. the table gradient is accumulated in half precision
// like the reference's; C == 1 (where the reference's at::Half atomicAdd is an empty stub) goes through a CAS on
// the aligned 32-bit word.
__device__ __forceinline__ void atomic_add_half(__half *dst, float v) {
    unsigned int *word = reinterpret_cast<unsigned int *>(reinterpret_cast<uintptr_t>(dst) & ~(uintptr_t)3);
    const bool upper = (reinterpret_cast<uintptr_t>(dst) & 2) != 0;
    unsigned int old = *word, assumed;
    do {
        assumed = old;
        const unsigned short cur = upper ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
        const __half sum = __float2half(__half2float(__ushort_as_half(cur)) + v);
        const unsigned int bits = __half_as_ushort(sum);
        const unsigned int repl = upper ? ((assumed & 0x0000ffffu) | (bits << 16)) : ((assumed & 0xffff0000u) | bits);
        old = atomicCAS(word, assumed, repl);
    } while (old != assumed);
}

template <typename T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_bwd_kernel(const T *__restrict__ grad, const float *__restrict__ inputs,
                                                       const int32_t *__restrict__ offsets,
                                                       T *__restrict__ grad_grid, uint32_t B, uint32_t L, float S,
                                                       uint32_t H, uint32_t gridtype, bool align_corners,
                                                       const GridLevels lv) {
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint64_t b = (uint64_t)blk * 256 + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    grad_grid += (uint64_t)(uint32_t)offsets[level] * C;
    const float *in = inputs + b * D;
    const T *g = grad + ((uint64_t)level * B + b) * C;

    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = in[d];
        if (x[d] < 0 || x[d] > 1) return;  // :252-257
    }
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    float scale;
    uint32_t resolution;
    level_params(lv, level, S, H, scale, resolution);
    const uint32_t dim_stride = align_corners ? resolution : resolution + 1;
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = mul_add_exact(x[d], scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    float gc[C];
    load_row<C>(g, gc);
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1.f;
        uint32_t pgl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) {
                w *= 1 - pos[d];
                pgl[d] = pg[d];
            } else {
                w *= pos[d];
                pgl[d] = pg[d] + 1;
            }
        }
        const uint32_t row = grid_row<D>(pgl, gridtype, dim_stride, hashmap_size);
        T *dst = grad_grid + (uint64_t)row * C;
        if constexpr (std::is_same<T, float>::value) {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) unsafeAtomicAdd(dst + c, w * gc[c]);
        } else if constexpr (C % 2 == 0) {
#pragma unroll
            for (uint32_t c = 0; c < C; c += 2)
                unsafeAtomicAdd(reinterpret_cast<__half2 *>(dst + c), __halves2half2(__float2half(w * gc[c]), __float2half(w * gc[c + 1])));
        } else {
            atomic_add_half(dst, __half2float(__float2half(w * gc[0])));
        }
    }
}

// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]   (:317-343)
// T = __half: `scalar_t result` of the reference is a half and every product / sum rounds to half (:331-340); the same
// sequence is evaluated here, so the result is bit-identical to a sequential half evaluation.
template <typename T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void grid_input_bwd_kernel(const T *__restrict__ grad,
                                                             const T *__restrict__ dy_dx,
                                                             T *__restrict__ grad_inputs, uint32_t B, uint32_t L) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)B * D) return;
    const uint64_t b = t / D;
    const uint32_t d = (uint32_t)(t - b * D);
    const T *dd = dy_dx + b * L * (uint64_t)(D * C);
    if constexpr (std::is_same<T, float>::value) {
        float r = 0.f;
        for (uint32_t l = 0; l < L; l++) {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) r += grad[((uint64_t)l * B + b) * C + c] * dd[(l * D + d) * C + c];
        }
        grad_inputs[t] = r;
    } else {
        __half r = __float2half(0.f);
        for (uint32_t l = 0; l < L; l++) {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                // at::Half arithmetic is "convert to float, operate, round back" (c10/util/Half-inl.h): the SUM of two
                // halfs is rounded to f32 first, then to half.  The empty asm keeps hipcc from contracting the chain into
                // native half instructions (v_add_f16 / v_fma_mix round once and differ by an ulp now and then).
                float prod = __half2float(grad[((uint64_t)l * B + b) * C + c]) * __half2float(dd[(l * D + d) * C + c]);
                float prod_h = __half2float(__float2half(prod));
                asm volatile("" : "+v"(prod_h));
                float sum = __half2float(r) + prod_h;
                asm volatile("" : "+v"(sum));
                r = __float2half(sum);
            }
        }
        grad_inputs[t] = r;
    }
}

template <typename T, uint32_t D>
int launch_fwd_c(const float *inputs, const T *emb, const int32_t *offsets, T *out, uint32_t B, uint32_t C, uint32_t L,
                 float S, uint32_t H, bool cg, T *dy_dx, uint32_t gridtype, bool ac, hipStream_t st) {
    const dim3 grid(sdn::div_up<uint32_t>(B, FWD_THREADS), L, 1);
    const dim3 block(FWD_THREADS);
    const GridLevels lv = make_levels(L, S, H);
    if constexpr (sizeof(T) == 4) {
        // inference form (f32 table, no dy_dx): the quad-cooperative kernel.  SDN_GRID_QUAD=0 keeps the one-lane-per-(sample,
        // level) kernel for A/B measurements.
        static const bool quad = [] { const char *e = getenv("SDN_GRID_QUAD"); return !(e && e[0] == '0'); }();
        if (quad && !cg) {
            const dim3 gq(sdn::div_up<uint32_t>(B, FWD_THREADS / 4), L, 1);
            const float *embf = (const float *)emb;
            float *outf = (float *)out;
            switch (C) {
                case 1: hipLaunchKernelGGL((grid_fwd_quad_kernel<D, 1>), gq, block, 0, st, inputs, embf, offsets, outf, B, L, S, H, gridtype, ac, lv); break;
                case 2: hipLaunchKernelGGL((grid_fwd_quad_kernel<D, 2>), gq, block, 0, st, inputs, embf, offsets, outf, B, L, S, H, gridtype, ac, lv); break;
                case 4: hipLaunchKernelGGL((grid_fwd_quad_kernel<D, 4>), gq, block, 0, st, inputs, embf, offsets, outf, B, L, S, H, gridtype, ac, lv); break;
                case 8: hipLaunchKernelGGL((grid_fwd_quad_kernel<D, 8>), gq, block, 0, st, inputs, embf, offsets, outf, B, L, S, H, gridtype, ac, lv); break;
                default: return sdn::fail(SDN_ERR_UNSUPPORTED, "GridEncoding: C must be 1, 2, 4, or 8.");
            }
            return sdn::check_launch("sdn_grid_encode_fwd");
        }
    }
    switch (C) {
        case 1: hipLaunchKernelGGL((grid_fwd_kernel<T, D, 1>), grid, block, 0, st, inputs, emb, offsets, out, B, L, S, H, cg, dy_dx, gridtype, ac, lv); break;
        case 2: hipLaunchKernelGGL((grid_fwd_kernel<T, D, 2>), grid, block, 0, st, inputs, emb, offsets, out, B, L, S, H, cg, dy_dx, gridtype, ac, lv); break;
        case 4: hipLaunchKernelGGL((grid_fwd_kernel<T, D, 4>), grid, block, 0, st, inputs, emb, offsets, out, B, L, S, H, cg, dy_dx, gridtype, ac, lv); break;
        case 8: hipLaunchKernelGGL((grid_fwd_kernel<T, D, 8>), grid, block, 0, st, inputs, emb, offsets, out, B, L, S, H, cg, dy_dx, gridtype, ac, lv); break;
        default: return sdn::fail(SDN_ERR_UNSUPPORTED, "GridEncoding: C must be 1, 2, 4, or 8.");
    }
    return sdn::check_launch("sdn_grid_encode_fwd");
}

template <typename T>
int launch_fwd(const float *inputs, const T *emb, const int32_t *offsets, T *out, uint32_t B, uint32_t D, uint32_t C,
               uint32_t L, float S, uint32_t H, bool cg, T *dy_dx, uint32_t gridtype, bool ac, hipStream_t st) {
    switch (D) {
        case 2: return launch_fwd_c<T, 2>(inputs, emb, offsets, out, B, C, L, S, H, cg, dy_dx, gridtype, ac, st);
        case 3: return launch_fwd_c<T, 3>(inputs, emb, offsets, out, B, C, L, S, H, cg, dy_dx, gridtype, ac, st);
        case 4: return launch_fwd_c<T, 4>(inputs, emb, offsets, out, B, C, L, S, H, cg, dy_dx, gridtype, ac, st);
        case 5: return launch_fwd_c<T, 5>(inputs, emb, offsets, out, B, C, L, S, H, cg, dy_dx, gridtype, ac, st);
        default: return sdn::fail(SDN_ERR_UNSUPPORTED, "GridEncoding: D must be 2, 3, 4, or 5.");
    }
}

template <typename T, uint32_t D, uint32_t C>
int launch_bwd_dc(const T *grad, const float *inputs, const int32_t *offsets, T *gg, uint32_t B, uint32_t L,
                  float S, uint32_t H, bool cg, const T *dy_dx, T *gi, uint32_t gridtype, bool ac,
                  hipStream_t st) {
    const dim3 grid(sdn::div_up<uint32_t>(B, 256), L, 1);
    hipLaunchKernelGGL((grid_bwd_kernel<T, D, C>), grid, dim3(256), 0, st, grad, inputs, offsets, gg, B, L, S, H, gridtype, ac,
                       make_levels(L, S, H));
    if (cg) {
        const uint64_t n = (uint64_t)B * D;
        hipLaunchKernelGGL((grid_input_bwd_kernel<T, D, C>), dim3((uint32_t)sdn::div_up<uint64_t>(n, 256)), dim3(256), 0,
                           st, grad, dy_dx, gi, B, L);
    }
    return sdn::check_launch("sdn_grid_encode_bwd");
}

template <typename T, uint32_t D>
int launch_bwd_d(const T *grad, const float *inputs, const int32_t *offsets, T *gg, uint32_t B, uint32_t C,
                 uint32_t L, float S, uint32_t H, bool cg, const T *dy_dx, T *gi, uint32_t gridtype, bool ac,
                 hipStream_t st) {
    switch (C) {
        case 1: return launch_bwd_dc<T, D, 1>(grad, inputs, offsets, gg, B, L, S, H, cg, dy_dx, gi, gridtype, ac, st);
        case 2: return launch_bwd_dc<T, D, 2>(grad, inputs, offsets, gg, B, L, S, H, cg, dy_dx, gi, gridtype, ac, st);
        case 4: return launch_bwd_dc<T, D, 4>(grad, inputs, offsets, gg, B, L, S, H, cg, dy_dx, gi, gridtype, ac, st);
        case 8: return launch_bwd_dc<T, D, 8>(grad, inputs, offsets, gg, B, L, S, H, cg, dy_dx, gi, gridtype, ac, st);
        default: return sdn::fail(SDN_ERR_UNSUPPORTED, "GridEncoding: C must be 1, 2, 4, or 8.");
    }
}

template <typename T>
int launch_bwd(const T *grad, const float *inputs, const int32_t *offsets, T *gg, uint32_t B, uint32_t D, uint32_t C,
               uint32_t L, float S, uint32_t H, bool cg, const T *dy_dx, T *gi, uint32_t gridtype, bool ac, hipStream_t st) {
    switch (D) {
        case 2: return launch_bwd_d<T, 2>(grad, inputs, offsets, gg, B, C, L, S, H, cg, dy_dx, gi, gridtype, ac, st);
        case 3: return launch_bwd_d<T, 3>(grad, inputs, offsets, gg, B, C, L, S, H, cg, dy_dx, gi, gridtype, ac, st);
        case 4: return launch_bwd_d<T, 4>(grad, inputs, offsets, gg, B, C, L, S, H, cg, dy_dx, gi, gridtype, ac, st);
        default: return launch_bwd_d<T, 5>(grad, inputs, offsets, gg, B, C, L, S, H, cg, dy_dx, gi, gridtype, ac, st);
    }
}

}  // namespace

extern "C" int sdn_grid_level_scales(uint32_t L, float S, uint32_t H, float *scales_host, uint32_t *resolutions_host) {
    SDN_REQUIRE(scales_host, "sdn_grid_level_scales: null pointer");
    for (uint32_t l = 0; l < L; l++) {
        const float sc = exp2f((float)l * S) * (float)H - 1.0f;  // gridencoder.cu:126
        scales_host[l] = sc;
        if (resolutions_host) resolutions_host[l] = (uint32_t)ceilf(sc) + 1;  // :127
    }
    return SDN_OK;
}

extern "C" int sdn_grid_encode_fwd(const float *inputs, const void *embeddings, int emb_dtype, const int32_t *offsets,
                                   void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   int calc_grad_inputs, void *dy_dx, uint32_t gridtype, int align_corners,
                                   sdn_stream_t stream) {
    if (D < 2 || D > 5) return sdn::fail(SDN_ERR_UNSUPPORTED, "GridEncoding: D must be 2, 3, 4, or 5.");
    if (!(C == 1 || C == 2 || C == 4 || C == 8))
        return sdn::fail(SDN_ERR_UNSUPPORTED, "GridEncoding: C must be 1, 2, 4, or 8.");
    SDN_REQUIRE(gridtype <= 1, "sdn_grid_encode_fwd: gridtype must be 0 (hash) or 1 (tiled)");
    SDN_REQUIRE(L >= 1 && L <= 65535, "sdn_grid_encode_fwd: L out of range");
    if (B == 0) return SDN_OK;
    SDN_REQUIRE(inputs && embeddings && offsets && outputs, "sdn_grid_encode_fwd: null pointer");
    SDN_REQUIRE(!calc_grad_inputs || dy_dx, "sdn_grid_encode_fwd: dy_dx required when calc_grad_inputs");
    hipStream_t st = (hipStream_t)stream;
    if (emb_dtype == SDN_F32)
        return launch_fwd<float>(inputs, (const float *)embeddings, offsets, (float *)outputs, B, D, C, L, S, H,
                                 calc_grad_inputs != 0, (float *)dy_dx, gridtype, align_corners != 0, st);
    if (emb_dtype == SDN_F16)
        return launch_fwd<__half>(inputs, (const __half *)embeddings, offsets, (__half *)outputs, B, D, C, L, S, H,
                                  calc_grad_inputs != 0, (__half *)dy_dx, gridtype, align_corners != 0, st);
    return sdn::fail(SDN_ERR_UNSUPPORTED, "sdn_grid_encode_fwd: embeddings must be f32 or f16");
}

extern "C" int sdn_grid_encode_bwd(const void *grad, const float *inputs, const void *embeddings, int emb_dtype,
                                   const int32_t *offsets, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                   uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void *dy_dx,
                                   void *grad_inputs, uint32_t gridtype, int align_corners, sdn_stream_t stream) {
    (void)embeddings;
    if (D < 2 || D > 5) return sdn::fail(SDN_ERR_UNSUPPORTED, "GridEncoding: D must be 2, 3, 4, or 5.");
    if (!(C == 1 || C == 2 || C == 4 || C == 8))
        return sdn::fail(SDN_ERR_UNSUPPORTED, "GridEncoding: C must be 1, 2, 4, or 8.");
    if (emb_dtype != SDN_F32 && emb_dtype != SDN_F16)
        return sdn::fail(SDN_ERR_UNSUPPORTED, "sdn_grid_encode_bwd: gradients must be f32 or f16");
    SDN_REQUIRE(gridtype <= 1, "sdn_grid_encode_bwd: gridtype must be 0 (hash) or 1 (tiled)");
    SDN_REQUIRE(L >= 1 && L <= 65535, "sdn_grid_encode_bwd: L out of range");
    if (B == 0) return SDN_OK;
    SDN_REQUIRE(grad && inputs && offsets && grad_embeddings, "sdn_grid_encode_bwd: null pointer");
    SDN_REQUIRE(!calc_grad_inputs || (dy_dx && grad_inputs), "sdn_grid_encode_bwd: dy_dx/grad_inputs required");
    hipStream_t st = (hipStream_t)stream;
    const bool cg = calc_grad_inputs != 0, ac = align_corners != 0;
    if (emb_dtype == SDN_F16)
        return launch_bwd<__half>((const __half *)grad, inputs, offsets, (__half *)grad_embeddings, B, D, C, L, S, H, cg,
                                  (const __half *)dy_dx, (__half *)grad_inputs, gridtype, ac, st);
    return launch_bwd<float>((const float *)grad, inputs, offsets, (float *)grad_embeddings, B, D, C, L, S, H, cg,
                             (const float *)dy_dx, (float *)grad_inputs, gridtype, ac, st);
}
