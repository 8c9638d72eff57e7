// The render CNN's convolutions on MFMA for gfx950 (RenderCNN, imaginaire/generators/gancraft_base.py:175-225): the 3x3
// 256 -> 256 ones (conv2a/2b/3a/3b) and, with TAPS = 1, the 1x1 ones (conv1 64 -> 256, conv4a/4b 256 -> 256; conv4
// 256 -> 3 + tanh is folded into conv4b's epilogue), with the same 3-term f16 split / f32 accumulate arithmetic
// as the field MLP (field.hip): the image must stay within 1e-3 of the fp32 reference, which plain f16 does not.
//
// Formulation (transposed implicit GEMM): D^T[cout][pixel] = sum_{tap, cin} W[cout][cin][tap] * X[cin][pixel + tap].
//   * MFMA columns = 32 pixels (an 8 x 4 patch) per wave, 8 waves (a 16 x 16 patch, 2 waves per SIMD) per workgroup;
//     all 256 output channels of the patch live in 128 accumulator registers;
//   * K = 9 taps x 256 channels = 144 k-steps of 16.  Per k-step a workgroup needs 16 KiB of weight fragments
//     (shared by its 4 waves, hi + lo) and every wave 2 KiB of its own activation fragments (hi + lo):
//     BOTH arrive by LDS-DMA.  For the activations the per-lane global address of global_load_lds is the
//     gather (pixel + tap, 8 consecutive channels = 16 B) and the lane-linear LDS image IS the MFMA B fragment;
//     activations are stored as two f16 planes laid out [16 channel chunks][padded pixel][16 channels] with a zero
//     border (taps need no branches): for one k-step (= one chunk) a wave's 32 pixels x 32 B are four contiguous
//     256-B runs.  (With channels-last [pixel][256] every 16-B piece sat in its own 128-B line: 8x L2->L1 read
//     amplification, the first version of this kernel was bound by that, not by the matrix pipe.)
//   * 5-slot LDS ring (32 KiB per slot = the whole 160 KiB LDS), 4 k-steps ahead, counted vmcnt + raw s_barrier per k-step exactly as
//     in field.hip; because every vector-memory operation in the main loop is a DMA with the same look-ahead, no
//     wait ever drains the pipeline (ordinary loads of B would: vmcnt completes in order);
//   * epilogue variants: bias + LeakyReLU -> f16 planes (conv2a/3a);  residual + style FiLM + LeakyReLU ->
//     fp32 rows and/or f16 planes (conv2b/3b, gancraft_base.py:197-200, :213-217).
// Precision (tools/precision_study.py): TERMS = 3 evaluates every product as Whi.Xhi + Wlo.Xhi + Whi.Xlo;
// TERMS = 1 (the four 3x3 convolutions by default) keeps Whi.Xhi only, with BOTH hi parts rounded to nearest
// (v_cvt_pk_f16_f32; round-toward-zero doubles the error and biases it).  Nothing downstream of the 3x3 layers
// amplifies their rounding noise, so the image moves by ~7e-5 rms / ~4e-4 max while the layer issues a third of
// the MFMAs and reads half the bytes.  In the 1-term layout a ring slot carries TWO k-steps (k0 in the "hi" position,
// k1 in the "lo" position of the 3-term layout): data movement per slot is unchanged, a unit is 4 MFMAs instead of 6.
// Roofline: MFMA (f16 dense peak); 3456 (TERMS = 3) / 1152 (TERMS = 1) MFMAs per wave per 32 pixels for a 3x3 layer.
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "sdn_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const char glb_char;

constexpr int CH = 256;
constexpr int A_BYTES = 16384;            // weight fragments of one k-step (8 row blocks, hi + lo)
constexpr int B_BYTES = 2048;             // one wave's activation fragments of one k-step (hi + lo)
constexpr int WAVES = 8;                  // 2 waves per SIMD: one wave's LDS/DMA/barrier time is the other's MFMA time
constexpr int SLOT_BYTES = A_BYTES + WAVES * B_BYTES;   // 32 KiB
constexpr int NSLOT = 5;                  // 160 KiB = the whole LDS of a CU
constexpr int AHEAD = 4;
constexpr int DMA_PER_SLOT = 4;           // per wave: 2 weight pieces + 2 activation pieces
constexpr int TILE_W = 8, TILE_H = 4;     // pixels per wave (MFMA columns)
constexpr int PATCH_W = 16, PATCH_H = 16; // pixels per workgroup: 2 x 4 wave tiles

struct ConvParams {
    const _Float16 *xh, *xl;   // input planes [16][Hb*Wb][16], zero border and zero outside the frame
    const char *wpk;           // packed weights, ksteps * 16 KiB
    int ksteps;                // ring slots per patch: TAPS * (input channels / 16) k-steps, two per slot when TERMS == 1
    const float *bias;         // [256] or nullptr
    const float *resid;        // fp32 [H*W][256] or nullptr
    const _Float16 *rh, *rl;   // residual as hi/lo planes (same layout as the output planes; may alias them) or nullptr
    const float *mod_w;        // [256] FiLM scale (applied as w + 1) or nullptr
    const float *mod_b;        // [256]
    _Float16 *oh, *ol;         // output planes or nullptr
    float *of32;               // fp32 [H*W][256] or nullptr
    const float *proj_w;       // [3][256] final 1x1 projection fused into the epilogue (conv4 + tanh), or nullptr
    const float *proj_b;       // [3]
    float *img;                // [3][H*W]
    int H, W, Hb, Wb;          // frame and padded-buffer extent (buffer pixel (y,x) -> (y+1, x+1))
    long chunk_bytes;          // Hb*Wb*32: byte stride between channel chunks of a plane
    unsigned row_inc, chunk_inc;   // 3x3 k-order increments of the activation offset: tap (r,2) -> (r+1,0), tap (2,2) -> (0,0) of the next chunk
    int gx, gy, n_groups;      // workgroup patches (16 x 16 pixels)
};

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// two f32 -> packed f16, round to nearest even: one v_cvt_pk_f16_f32 (new in gfx950)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2v cvt_rtn(float a, float b) {
    return __builtin_convertvector(float2v{a, b}, half2v);
}

// ---- the DMA stream ----------------------------------------------------------------------------------------------------
// k order is channel-chunk major: k = 9*s + tap.  The 9 taps of one 16-channel chunk re-read the same (patch + halo) x 32 B
// region, which stays in L1/L2; with tap-major order every tap re-streamed the whole patch from the Infinity Cache (measured
// 3.2 GB fetched per launch for 0.58 GB of input).
// The stream's position is kept INCREMENTALLY in scalar registers (a byte offset inside a plane, measured from tap (0,0) of
// chunk 0, a tap counter and the ring-slot index of the weights): the first version recomputed k / 9, k % 9, tap / 3 and a
// 64-bit multiply-add per activation piece, ~30 scalar + 6 vector instructions each, as many issue slots as the slot's MFMAs.
// Every DMA's address is (scalar base) + (32-bit per-lane offset), the saddr form of global_load_lds: no vector address arithmetic.
struct Stream {
    unsigned long off;   // activation byte offset of the current k-step
    int tap;             // 0..8 (3x3 only)
    int w;               // ring slot inside the patch (weights)
};

template <int TAPS>
__device__ __forceinline__ void stream_advance_k(const ConvParams &p, Stream &st) {
    if constexpr (TAPS == 9) {
        unsigned inc = ((0x24u >> st.tap) & 1u) ? p.row_inc : 32u;   // taps 2 and 5 end a row
        inc = st.tap == 8 ? p.chunk_inc : inc;
        st.off += inc;
        st.tap = st.tap == 8 ? 0 : st.tap + 1;
    } else {
        st.off += (unsigned long)p.chunk_bytes;   // 1x1: k-step = channel chunk
    }
}

// scalar operands of the 4 DMA pieces a wave contributes to one ring slot: 0, 1 = its 2 KiB of the weight fragments,
// 2 / 3 = its own activation fragments (TERMS == 3: hi / lo plane of one k-step; TERMS == 1: hi plane of two k-steps)
struct SlotIssue {
    const char *w;        // weights of the slot + this wave's 2 KiB
    const char *a0, *a1;  // plane base + k-step offset of piece 2 / 3
    unsigned voff;        // per-lane activation offset (this patch's or the next one's)
    char *dst;            // ring position
};

template <int TAPS, int TERMS>
__device__ __forceinline__ SlotIssue stream_next_slot(const ConvParams &p, Stream &st, char *lds, int pos, int wave, unsigned voff, int ksteps) {
    SlotIssue si;
    si.w = p.wpk + (size_t)st.w * A_BYTES + wave * 2048;
    si.a0 = (const char *)p.xh + st.off;
    if constexpr (TERMS == 1) {
        stream_advance_k<TAPS>(p, st);
        si.a1 = (const char *)p.xh + st.off;
    } else {
        si.a1 = (const char *)p.xl + st.off;
    }
    stream_advance_k<TAPS>(p, st);
    st.w++;
    if (st.w == ksteps) {   // the stream runs on into the next patch
        st.w = 0;
        st.off = 0;
    }
    si.voff = voff;
    si.dst = lds + pos * SLOT_BYTES;
    return si;
}

// The same, a quarter at a time: the main loop computes the operands of the NEXT slot's pieces in the shadow of this slot's
// MFMAs (chunk U in unit U).  As one block at the top of a slot these ~45 scalar instructions were matrix-pipe idle time:
// both waves of a SIMD run in lockstep behind the per-slot barrier, so neither had MFMAs in flight meanwhile.
// The empty asm statements pin a chunk to its gap (inputs not earlier, outputs not later).
template <int TAPS, int TERMS, int U>
__device__ __forceinline__ void stream_chunk(const ConvParams &p, Stream &st, SlotIssue &sn, char *lds, int &pos, int wave,
                                             unsigned voff_sel, int ksteps) {
    if constexpr (U == 0) {
        asm volatile("" : "+s"(st.w));
        sn.w = p.wpk + (size_t)st.w * A_BYTES + wave * 2048;
        sn.dst = lds + pos * SLOT_BYTES;
        pos = pos + 1 == NSLOT ? 0 : pos + 1;
        asm volatile("" ::"s"(sn.w), "s"(pos));
    } else if constexpr (U == 1) {
        asm volatile("" : "+s"(st.tap));
        sn.a0 = (const char *)p.xh + st.off;
        if constexpr (TERMS == 1) stream_advance_k<TAPS>(p, st);
        asm volatile("" ::"s"(sn.a0), "s"(st.off), "s"(st.tap));
    } else if constexpr (U == 2) {
        asm volatile("" : "+s"(st.tap));
        sn.a1 = (const char *)(TERMS == 1 ? p.xh : p.xl) + st.off;
        stream_advance_k<TAPS>(p, st);
        asm volatile("" ::"s"(sn.a1), "s"(st.off), "s"(st.tap));
    } else {
        asm volatile("" : "+s"(st.w));
        st.w++;
        if (st.w == ksteps) {   // the stream runs on into the next patch
            st.w = 0;
            st.off = 0;
        }
        asm volatile("" ::"s"(st.w), "s"(st.off));
    }
}

template <int PIECE>
__device__ __forceinline__ void issue_piece(const SlotIssue &si, int wave, int lane) {
    // (the instruction offset applies to the global AND the LDS address)
    if constexpr (PIECE == 0) __builtin_amdgcn_global_load_lds((glb_char *)(si.w + (unsigned)(lane * 16)), (lds_char *)(si.dst + wave * 2048), 16, 0, 0);
    if constexpr (PIECE == 1) __builtin_amdgcn_global_load_lds((glb_char *)(si.w + (unsigned)(lane * 16)), (lds_char *)(si.dst + wave * 2048), 16, 1024, 0);
    if constexpr (PIECE == 2) __builtin_amdgcn_global_load_lds((glb_char *)(si.a0 + si.voff), (lds_char *)(si.dst + A_BYTES + wave * B_BYTES), 16, 0, 0);
    if constexpr (PIECE == 3) __builtin_amdgcn_global_load_lds((glb_char *)(si.a1 + si.voff), (lds_char *)(si.dst + A_BYTES + wave * B_BYTES + 1024), 16, 0, 0);
}

// Fragment reads are inline asm with hand-counted s_waitcnt: behind a pending LDS-DMA the compiler's own wait
// insertion degrades every LDS wait to lgkmcnt(0), which drains the prefetch issued just before it and exposes one
// LDS round trip per 6 MFMAs (measured: 1.80 -> see DESIGN.md per 3x3 launch).
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p;
}

template <int OFF>
__device__ __forceinline__ void ds_read16(half8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

template <int U>
__device__ __forceinline__ void lds_unit(unsigned slot_lane, half8 (&a)[4]) {
    ds_read16<U * 4096>(a[0], slot_lane);
    ds_read16<U * 4096 + 1024>(a[1], slot_lane);
    ds_read16<U * 4096 + 2048>(a[2], slot_lane);
    ds_read16<U * 4096 + 3072>(a[3], slot_lane);
}

template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// One unit (2 row blocks x one k-step, 6 MFMAs).  Behind each MFMA goes ONE piece of other work, so that the
// texture path (DMA), the LDS and the matrix pipe all stay busy: a DMA piece of the slot AHEAD k-steps on, the
// activation fragments of the next k-step (U == 2) and the weight fragments of the unit two units on.  The fragment
// reads are the last LDS operations of a unit: "lgkmcnt(reads of the previous unit)" at the start of a unit means
// this unit's fragments have landed.
template <int TAPS, int TERMS, int DBG, int U>
__device__ __forceinline__ void conv_unit(f32x16 (&acc)[8], half8 (&a)[4][4], half8 (&bcur)[2], half8 (&bnext)[2], unsigned slot,
                                          unsigned slot_n, unsigned b_off, const SlotIssue &si, int wave, int lane,
                                          const ConvParams &p, Stream &st, SlotIssue &sn, char *lds, int &pos_issue, unsigned voff_sel, int ksteps) {
    constexpr int ib = 2 * U;
    half8(&au)[4] = a[U];
    half8(&nx)[4] = a[(U + 2) & 3];
    const unsigned src = U < 2 ? slot : slot_n;            // unit U+2 of this slot, or unit U-2 of the next one
    constexpr int OFF = ((U + 2) & 3) * 4096;
    if constexpr (!(DBG & 64)) lds_wait<U == 3 ? 6 : 4>();
#define SDN_GAP(K) \
    if constexpr (K == 0 && !(DBG & 1)) issue_piece<U>(si, wave, lane); \
    if constexpr (K == 0 && U == 2 && !(DBG & 64)) { ds_read16<0>(bnext[0], slot_n + b_off); ds_read16<1024>(bnext[1], slot_n + b_off); } \
    if constexpr (K >= 1 && K <= 4 && !(DBG & 64)) ds_read16<OFF + (K - 1) * 1024>(nx[K - 1], src); \
    if constexpr (K == 2) stream_chunk<TAPS, TERMS, U>(p, st, sn, lds, pos_issue, wave, voff_sel, ksteps); \
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG & 16) {
        asm volatile("" ::"v"(au[0]), "v"(au[1]), "v"(au[2]), "v"(au[3]), "v"(bcur[0]), "v"(bcur[1]));
        SDN_GAP(0) SDN_GAP(1) SDN_GAP(2) SDN_GAP(3) SDN_GAP(4)
    } else if constexpr (TERMS == 1) {
        // fragments: au[0] = (ib, k0), au[1] = (ib, k1), au[2] = (ib+1, k0), au[3] = (ib+1, k1); bcur[0] = k0, bcur[1] = k1
        acc[ib] = mfma16(au[0], bcur[0], acc[ib]);
        __builtin_amdgcn_sched_barrier(0);
        SDN_GAP(0)
        acc[ib + 1] = mfma16(au[2], bcur[0], acc[ib + 1]);
        SDN_GAP(1) SDN_GAP(2)
        acc[ib] = mfma16(au[1], bcur[1], acc[ib]);
        SDN_GAP(3)
        acc[ib + 1] = mfma16(au[3], bcur[1], acc[ib + 1]);
        SDN_GAP(4)
    } else {
        acc[ib] = mfma16(au[0], bcur[0], acc[ib]);
        __builtin_amdgcn_sched_barrier(0);   // the matrix instruction first: the gap's address arithmetic runs in its shadow
        SDN_GAP(0)
        acc[ib + 1] = mfma16(au[2], bcur[0], acc[ib + 1]);
        SDN_GAP(1)
        acc[ib] = mfma16(au[1], bcur[0], acc[ib]);
        SDN_GAP(2)
        acc[ib + 1] = mfma16(au[3], bcur[0], acc[ib + 1]);
        SDN_GAP(3)
        acc[ib] = mfma16(au[0], bcur[1], acc[ib]);
        SDN_GAP(4)
        acc[ib + 1] = mfma16(au[2], bcur[1], acc[ib + 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef SDN_GAP
}

__device__ __forceinline__ long lane_pixel_offset(const ConvParams &p, int grp, int wave, int lane, int &py, int &px) {
    const int gyi = grp / p.gx, gxi = grp - gyi * p.gx;
    const int j = lane & 31, h = lane >> 5;
    py = gyi * PATCH_H + (wave >> 1) * TILE_H + (j >> 3);
    px = gxi * PATCH_W + (wave & 1) * TILE_W + (j & 7);
    return ((long)(py + 1) * p.Wb + (px + 1)) * 32 + h * 16;   // byte offset inside chunk 0
}

// EPI: what the epilogue reads and writes besides (bias ->) LeakyReLU -> hi plane, known at compile time in the variants the
// render CNN uses (straight-line code: the loads of several iterations overlap): 1 residual planes, 2 FiLM, 8 NO bias,
// 16 lo plane too; 255 = anything, decided at run time from the pointers (incl. the fused projection, fp32 residual / output).
template <int TAPS, int DBG, int TERMS, int EPI>
__global__ __launch_bounds__(512, 2) void conv_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[NSLOT * SLOT_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    int grp = blockIdx.x;
    if (grp >= p.n_groups) return;
    // per-lane byte offset of the lane's pixel inside chunk 0, measured from the first tap of the k order (3x3: the pixel
    // one row up and one column left; the zero border makes that address valid for every pixel of the frame)
    const unsigned tap0 = TAPS == 9 ? (unsigned)(p.Wb + 1) * 32u : 0u;
    int py, px;
    unsigned voff = (unsigned)lane_pixel_offset(p, grp, wave, lane, py, px) - tap0;
    int grp_n = grp + gridDim.x;
    int pyn, pxn;
    unsigned voff_n = grp_n < p.n_groups ? (unsigned)lane_pixel_offset(p, grp_n, wave, lane, pyn, pxn) - tap0 : voff;

    const int ksteps = p.ksteps;
    // De-phase the workgroups.  All of them start together and take the same time per patch, so their epilogues coincide:
    // 256 x 128 KiB (hi plane) leave the chip in one burst every patch time, and the stores back up behind the fabric.
    // Workgroups that have one patch less than the longest-running ones (n_groups mod grid of them have one more) start
    // late by a fraction of a patch time: it costs nothing (they still finish first) and spreads the bursts.
    {
        const int n_mine = (p.n_groups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
        const int n_max = (p.n_groups + (int)gridDim.x - 1) / (int)gridDim.x;
        if (!(DBG & 2048) && n_mine < n_max) {
            const int phase = (blockIdx.x >> 3) & 15;                    // blockIdx & 7 is the XCD
            const int naps = (phase * (ksteps * 1400 + 20000)) >> (4 + 13);   // s_sleep 127 ~ 8 k cycles
            for (int i = 0; i < naps; i++) __builtin_amdgcn_s_sleep(127);
        }
    }
    Stream st{0ul, 0, 0};
    int pos_issue = 0, pos_use = 0;
#pragma unroll
    for (int q = 0; q < AHEAD; q++) {
        const SlotIssue si = stream_next_slot<TAPS, TERMS>(p, st, lds, pos_issue, wave, voff, ksteps);
        issue_piece<0>(si, wave, lane);
        issue_piece<1>(si, wave, lane);
        issue_piece<2>(si, wave, lane);
        issue_piece<3>(si, wave, lane);
        pos_issue = pos_issue + 1 == NSLOT ? 0 : pos_issue + 1;
    }
    // operands of the pieces issued during the first slot of the loop; from then on each slot prepares the next one's
    SlotIssue si = stream_next_slot<TAPS, TERMS>(p, st, lds, pos_issue, wave, AHEAD < ksteps ? voff : voff_n, ksteps);
    pos_issue = pos_issue + 1 == NSLOT ? 0 : pos_issue + 1;

    // Fragment registers: the weight fragments of 4 units (the unit in use, the next one, the one being read) and two
    // sets of this wave's activation fragments (k-steps alternate between them: the loop is unrolled by two so that
    // no register copies are needed).  128 accumulators + 64 + 16 stay below the 256 registers of 2 waves / SIMD.
    half8 a[4][4];
    half8 b[2][2];
    const unsigned lds_lane = lds_addr(lds) + lane * 16;
    const unsigned b_off = A_BYTES + wave * B_BYTES;
    bool first_patch = true;
    // fragments of the kernel's very first slot (from then on every slot's first fragments are read during the slot before it)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * DMA_PER_SLOT) : "memory");
    __builtin_amdgcn_s_barrier();
    ds_read16<0>(b[0][0], lds_lane + b_off);
    ds_read16<1024>(b[0][1], lds_lane + b_off);
    lds_unit<0>(lds_lane, a[0]);
    lds_unit<1>(lds_lane, a[1]);

    while (true) {
        f32x16 acc[8];
#pragma unroll
        for (int ib = 0; ib < 8; ib++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[ib][r] = 0.f;
#pragma unroll 1
        for (int kt2 = 0; kt2 < ksteps; kt2 += 2) {
#pragma unroll
            for (int par = 0; par < 2; par++) {
                const int kt = kt2 + par;
                // `si`: the slot fetched during this one (AHEAD slots on, possibly of the next patch); `sn`: the one after it
                // (the per-lane pixel offsets are chosen HERE, not a slot ahead: with ksteps == AHEAD, the 64-channel conv1, the
                // last slot of a patch prepares a slot of the patch after the next one, whose offsets are not known before
                // the epilogue)
                SlotIssue sn;
                si.voff = kt + AHEAD < ksteps ? voff : voff_n;
                const unsigned voff_sel = si.voff;
                const unsigned slot = lds_lane + pos_use * SLOT_BYTES;
                pos_use = pos_use + 1 == NSLOT ? 0 : pos_use + 1;
                const unsigned slot_n = lds_lane + pos_use * SLOT_BYTES;
                asm volatile("" ::"v"(voff_sel), "v"(slot), "v"(slot_n));   // ... computed HERE, before the barrier
                __builtin_amdgcn_sched_barrier(0);
                // ---- acquire slot kt: mine of slots kt and kt+1 have landed, then everybody's; slot kt-1 is free ---
                // (not in the first AHEAD - 1 slots behind an epilogue: those slots were drained in front of it, and its stores
                // count in vmcnt too -- vector memory operations retire in order, so this wait would expose their whole write
                // latency at the start of every patch; by slot AHEAD - 1 they have retired)
                if (kt >= AHEAD - 1 || first_patch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 2) * DMA_PER_SLOT) : "memory");
                if constexpr (!(DBG & 2)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                conv_unit<TAPS, TERMS, DBG, 0>(acc, a, b[par], b[par ^ 1], slot, slot_n, b_off, si, wave, lane, p, st, sn, lds, pos_issue, voff_sel, ksteps);
                conv_unit<TAPS, TERMS, DBG, 1>(acc, a, b[par], b[par ^ 1], slot, slot_n, b_off, si, wave, lane, p, st, sn, lds, pos_issue, voff_sel, ksteps);
                conv_unit<TAPS, TERMS, DBG, 2>(acc, a, b[par], b[par ^ 1], slot, slot_n, b_off, si, wave, lane, p, st, sn, lds, pos_issue, voff_sel, ksteps);
                conv_unit<TAPS, TERMS, DBG, 3>(acc, a, b[par], b[par ^ 1], slot, slot_n, b_off, si, wave, lane, p, st, sn, lds, pos_issue, voff_sel, ksteps);
                si = sn;
            }
        }
        // the prefetches of the (possibly non-existent) next patch's first units must land before registers are reused
        // (and the next patch's first AHEAD slots, all in flight by now, before the epilogue's stores are issued)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        first_patch = false;

        // ---- epilogue of this patch -----------------------------------------------------------------------------
        const bool in_frame = py < p.H && px < p.W;
        if (in_frame) {
            const long orow = (long)py * p.W + px;                          // fp32 rows are unpadded
            const long ppix = (long)(py + 1) * p.Wb + (px + 1);             // padded pixel index
            float pj[3] = {0.f, 0.f, 0.f};
            const long plane_px = (long)p.Hb * p.Wb;
            // The packed weights permute the rows of every 32-channel block (pack_conv_kernel) so that accumulator
            // register r of half-wave h is channel 32*ib + 16*h + r: a lane owns one whole 16-channel chunk of its pixel
            // per row block = 32 contiguous bytes of each plane, and the 8 pixels of a tile row are 256 contiguous bytes.
            //
            // TWO passes over the accumulators.  Pass 1 loads (bias, residual, FiLM, projection rows) and leaves the activation
            // in the accumulator registers; pass 2 converts and stores.  Vector memory operations retire in order, so in
            // the one-pass form (load, compute, store per 8 channels) every iteration's wait for its loads was also a wait for
            // the previous iteration's stores to be acknowledged: 16 exposed store round trips per patch, ~24 k cycles of a
            // ~120 k-cycle patch of the 1-term 3x3 layers (cycle counters) and most of a 1x1 layer's time.
            constexpr bool SPEC = EPI != 255;
            const bool has_bias = SPEC ? !(EPI & 8) : p.bias != nullptr, has_rp = SPEC ? bool(EPI & 1) : p.rh != nullptr;
            const bool has_mod = SPEC ? bool(EPI & 2) : p.mod_w != nullptr, has_r32 = !SPEC && p.resid;
            const bool has_proj = !SPEC && p.proj_w, has_o32 = !SPEC && p.of32;
            const bool has_oh = SPEC || p.oh, has_ol = SPEC ? bool(EPI & 16) : p.ol != nullptr;
            int he = h;   // opaque per patch: otherwise the per-channel addresses below are hoisted out of the patch loop and
            asm volatile("" : "+v"(he));   // their registers stay live across the main loop
#pragma unroll
            for (int ib = 0; ib < 8; ib++) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int c0 = 32 * ib + 16 * he + 8 * q;                // 8 consecutive channels
                    const long po = ((long)(2 * ib + he) * plane_px + ppix) * 16 + 8 * q;   // element offset inside a plane
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = acc[ib][8 * q + e];
                    if (has_bias) {
                        const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + c0), b1 = *reinterpret_cast<const float4 *>(p.bias + c0 + 4);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    }
                    if (has_r32) {   // y = y + conv(...)   (gancraft_base.py:213, :216)
                        const float4 r0 = *reinterpret_cast<const float4 *>(p.resid + orow * CH + c0);
                        const float4 r1 = *reinterpret_cast<const float4 *>(p.resid + orow * CH + c0 + 4);
                        v[0] = r0.x + v[0]; v[1] = r0.y + v[1]; v[2] = r0.z + v[2]; v[3] = r0.w + v[3];
                        v[4] = r1.x + v[4]; v[5] = r1.y + v[5]; v[6] = r1.z + v[6]; v[7] = r1.w + v[7];
                    } else if (has_rp) {   // the same, y kept as hi + lo planes (exact to 2^-22; read before the in-place store)
                        const half8 h8 = *reinterpret_cast<const half8 *>(p.rh + po);
                        const half8 l8 = *reinterpret_cast<const half8 *>(p.rl + po);
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] = ((float)h8[e] + (float)l8[e]) + v[e];
                    }
                    if (has_mod) {   // modulate: x * (w + 1) + b   (:197-200)
#pragma unroll
                        for (int e4 = 0; e4 < 2; e4++) {
                            const float4 mw = *reinterpret_cast<const float4 *>(p.mod_w + c0 + 4 * e4);
                            const float4 mb = *reinterpret_cast<const float4 *>(p.mod_b + c0 + 4 * e4);
                            v[4 * e4] = v[4 * e4] * (mw.x + 1.f) + mb.x; v[4 * e4 + 1] = v[4 * e4 + 1] * (mw.y + 1.f) + mb.y;
                            v[4 * e4 + 2] = v[4 * e4 + 2] * (mw.z + 1.f) + mb.z; v[4 * e4 + 3] = v[4 * e4 + 3] * (mw.w + 1.f) + mb.w;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = vmax(v[e], 0.2f * v[e]);   // LeakyReLU(0.2)
                    if (has_proj) {   // conv4 (256 -> 3, 1x1): this lane's 8 channels of its pixel
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const float4 w0 = *reinterpret_cast<const float4 *>(p.proj_w + c * CH + c0);
                            const float4 w1 = *reinterpret_cast<const float4 *>(p.proj_w + c * CH + c0 + 4);
                            pj[c] += w0.x * v[0] + w0.y * v[1] + w0.z * v[2] + w0.w * v[3] + w1.x * v[4] + w1.y * v[5] + w1.z * v[6] + w1.w * v[7];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) acc[ib][8 * q + e] = v[e];
                }
            }
            // ---- pass 2: stores only
            if (has_o32 || has_oh) {
#pragma unroll
                for (int ib = 0; ib < 8; ib++) {
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int c0 = 32 * ib + 16 * he + 8 * q;
                        const long po = ((long)(2 * ib + he) * plane_px + ppix) * 16 + 8 * q;
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] = acc[ib][8 * q + e];
                        if (has_o32) {
                            *reinterpret_cast<float4 *>(p.of32 + orow * CH + c0) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4 *>(p.of32 + orow * CH + c0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        }
                        if (has_oh) {
                            // hi = round-to-nearest f16 (v_cvt_pk_f16_f32): a 1-term consumer sees half the error of a
                            // truncated hi, a 3-term consumer does not care (lo absorbs the remainder either way)
                            half8 hv, lv;
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                const half2v hp = cvt_rtn(v[e], v[e + 1]);
                                hv[e] = hp[0]; hv[e + 1] = hp[1];
                                const half2v lp = cvt_rtn(v[e] - (float)hp[0], v[e + 1] - (float)hp[1]);
                                lv[e] = lp[0]; lv[e + 1] = lp[1];
                            }
                            *reinterpret_cast<half8 *>(p.oh + po) = hv;
                            if (has_ol) *reinterpret_cast<half8 *>(p.ol + po) = lv;   // not needed when every consumer is 1-term
                        }
                    }
                }
            }
            if (has_proj) {   // the two half-waves hold the two channel halves of the same pixel
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float t = pj[c] + __shfl_xor(pj[c], 32);
                    if (h == 0) p.img[(long)c * p.H * p.W + orow] = tanhf(t + p.proj_b[c]);   // gancraft_base.py:603
                }
            }
        }
        // ---- next patch ---------------------------------------------------------------------------------------------
        grp = grp_n;
        if (grp >= p.n_groups) break;
        voff = voff_n;
        py = pyn;
        px = pxn;
        grp_n = grp + gridDim.x;
        voff_n = grp_n < p.n_groups ? (unsigned)lane_pixel_offset(p, grp_n, wave, lane, pyn, pxn) - tap0 : voff;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// ---- weight packing: W [256][cin][taps] (PyTorch OIHW) -> [k-step t = taps*s + tap][unit][frag][64 lanes][8] ----------
__global__ __launch_bounds__(256) void pack_conv_kernel(const float *__restrict__ W, half8 *__restrict__ out, int cin, int taps, int terms) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;   // (t, ib, lane), t = ring slot
    const int nslots = (cin / 16) * taps / (terms == 1 ? 2 : 1);
    if (g >= (size_t)nslots * 8 * 64) return;
    const int lane = (int)(g % 64);
    const int ib = (int)((g / 64) % 8);
    const int t = (int)(g / (64 * 8));
    // MFMA row i of a 32-row block carries channel 16*((i>>2)&1) + 4*(i>>3) + (i&3) of the block: the C/D register
    // layout (row = (r&3) + 8*(r>>2) + 4*h) then gives lane half h the 16 CONSECUTIVE channels 16*h + r (epilogue)
    const int i32 = lane & 31, h = lane >> 5;
    const int co = 32 * ib + 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);
    half8 hi, lo;
    if (terms == 1) {   // slot t = k-steps 2t ("hi" position) and 2t+1 ("lo" position), round-to-nearest f16 of the weight
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int k = 2 * t + kk, s = k / taps, tap = k - taps * s;   // same k order as tap_offset
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const _Float16 vh = (_Float16)W[((size_t)co * cin + 16 * s + 8 * h + e) * taps + tap];
                if (kk == 0) hi[e] = vh;
                else lo[e] = vh;
            }
        }
    } else {
        const int s = t / taps, tap = t - taps * s;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int ci = 16 * s + 8 * h + e;
            const float v = W[((size_t)co * cin + ci) * taps + tap];
            const _Float16 vh = (_Float16)v;
            hi[e] = vh;
            lo[e] = (_Float16)(v - (float)vh);
        }
    }
    // unit = ib / 2; fragments (ib,hi) (ib,lo) (ib+1,hi) (ib+1,lo)   [1-term: (ib,k0) (ib,k1) (ib+1,k0) (ib+1,k1)]
    const size_t base = ((size_t)t * 16 + (size_t)(ib / 2) * 4 + (ib & 1) * 2) * 64 + lane;
    out[base] = hi;
    out[base + 64] = lo;
}

// ---- fp32 rows [H*W][C] -> padded f16 hi / lo planes [C/16][Hb*Wb][16] ------------------------------------------------
__global__ __launch_bounds__(256) void planes_kernel(const float *__restrict__ x, _Float16 *__restrict__ oh,
                                                     _Float16 *__restrict__ ol, int H, int W, int Hb, int Wb, int C) {
    const long n = (long)H * W * (C / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long pix = i / (C / 4);
        const int c0 = (int)(i % (C / 4)) * 4;
        const int y = (int)(pix / W), xx = (int)(pix % W);
        const float4 v = *reinterpret_cast<const float4 *>(x + pix * C + c0);
        const half2v h0 = cvt_rtn(v.x, v.y), h1 = cvt_rtn(v.z, v.w);
        const half2v l0 = cvt_rtn(v.x - (float)h0[0], v.y - (float)h0[1]);
        const half2v l1 = cvt_rtn(v.z - (float)h1[0], v.w - (float)h1[1]);
        half4 hv, lv;
        hv[0] = h0[0]; hv[1] = h0[1]; hv[2] = h1[0]; hv[3] = h1[1];
        lv[0] = l0[0]; lv[1] = l0[1]; lv[2] = l1[0]; lv[3] = l1[1];
        const long o = ((long)(c0 >> 4) * ((long)Hb * Wb) + (long)(y + 1) * Wb + (xx + 1)) * 16 + (c0 & 15);
        *reinterpret_cast<half4 *>(oh + o) = hv;
        *reinterpret_cast<half4 *>(ol + o) = lv;
    }
}

}  // namespace

extern "C" {

// padded plane extent for an H x W frame: multiples of the 16 x 16 workgroup patch + 1-pixel zero border
void sdn_conv_plane_dims(int H, int W, int *Hb, int *Wb) {
    *Hb = sdn::div_up(H, PATCH_H) * PATCH_H + 2;
    *Wb = sdn::div_up(W, PATCH_W) * PATCH_W + 2;
}

static bool conv_shape_ok(int cin, int taps, int terms) {
    if (terms != 1 && terms != 3) return false;
    if (terms == 1 && !(taps == 9 && cin == 256)) return false;   // the 1-term kernel exists for the 3x3 layers
    return (taps == 9 && cin == 256) || (taps == 1 && cin >= 64 && cin <= 256 && cin % 16 == 0);
}
#define SDN_CONV_SHAPES "supported: 3x3 256->256 (terms 1 or 3) and 1x1 (64..256, multiple of 16)->256 (terms 3)"

static int conv_slots(int cin, int taps, int terms) { return (cin / 16) * taps / (terms == 1 ? 2 : 1); }

size_t sdn_conv_packed_weight_bytes(int cin, int taps, int terms) {
    return conv_shape_ok(cin, taps, terms) ? (size_t)conv_slots(cin, taps, terms) * A_BYTES : 0;
}

int sdn_conv_pack_weights(const float *w_oihw, int cin, int taps, int terms, void *packed, sdn_stream_t stream) {
    SDN_REQUIRE(w_oihw && packed, "sdn_conv_pack_weights: null pointer");
    SDN_REQUIRE(conv_shape_ok(cin, taps, terms), "sdn_conv_pack_weights: " SDN_CONV_SHAPES);
    const size_t n = (size_t)conv_slots(cin, taps, terms) * 8 * 64;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)sdn::div_up<size_t>(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, (half8 *)packed, cin, taps, terms);
    return sdn::check_launch("sdn_conv_pack_weights");
}

int sdn_conv_planes_from_f32(const float *x, int channels, void *out_hi, void *out_lo, int H, int W, sdn_stream_t stream) {
    SDN_REQUIRE(x && out_hi && out_lo && H > 0 && W > 0, "sdn_conv_planes_from_f32: bad argument");
    SDN_REQUIRE(channels >= 16 && channels <= 256 && channels % 16 == 0, "sdn_conv_planes_from_f32: channels must be a multiple of 16 in [16, 256]");
    int Hb, Wb;
    sdn_conv_plane_dims(H, W, &Hb, &Wb);
    hipLaunchKernelGGL(planes_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, x, (_Float16 *)out_hi, (_Float16 *)out_lo,
                       H, W, Hb, Wb, channels);
    return sdn::check_launch("sdn_conv_planes_from_f32");
}

int sdn_conv(const void *in_hi, const void *in_lo, int cin, int taps, int terms, const void *packed, const float *bias, const float *resid,
             const void *resid_hi, const void *resid_lo, const float *mod_w, const float *mod_b, void *out_hi, void *out_lo,
             float *out_f32, const float *proj_w, const float *proj_b, float *out_img, int H, int W, int n_workgroups,
             sdn_stream_t stream) {
    SDN_REQUIRE(in_hi && packed && H > 0 && W > 0, "sdn_conv: bad argument");
    SDN_REQUIRE(conv_shape_ok(cin, taps, terms), "sdn_conv: " SDN_CONV_SHAPES);
    SDN_REQUIRE(in_lo || terms == 1, "sdn_conv: the 3-term product needs the lo input plane");
    SDN_REQUIRE(out_hi || out_f32 || out_img, "sdn_conv: no output requested");
    SDN_REQUIRE(out_hi || !out_lo, "sdn_conv: out_lo without out_hi");
    SDN_REQUIRE((mod_w == nullptr) == (mod_b == nullptr), "sdn_conv: mod_w and mod_b go together");
    SDN_REQUIRE((resid_hi == nullptr) == (resid_lo == nullptr) && !(resid && resid_hi), "sdn_conv: one residual form at most");
    SDN_REQUIRE(!(resid_hi && (resid_hi == in_hi || resid_lo == in_lo)), "sdn_conv: the residual planes must not be the input planes");
    SDN_REQUIRE((proj_w == nullptr) == (out_img == nullptr) && (proj_w == nullptr) == (proj_b == nullptr),
                "sdn_conv: proj_w, proj_b and out_img go together");
    ConvParams p;
    p.xh = (const _Float16 *)in_hi; p.xl = (const _Float16 *)in_lo; p.wpk = (const char *)packed;
    p.ksteps = conv_slots(cin, taps, terms);
    p.bias = bias; p.resid = resid; p.rh = (const _Float16 *)resid_hi; p.rl = (const _Float16 *)resid_lo; p.mod_w = mod_w; p.mod_b = mod_b;
    p.oh = (_Float16 *)out_hi; p.ol = (_Float16 *)out_lo; p.of32 = out_f32;
    p.proj_w = proj_w; p.proj_b = proj_b; p.img = out_img;
    p.H = H; p.W = W;
    sdn_conv_plane_dims(H, W, &p.Hb, &p.Wb);
    p.chunk_bytes = (long)p.Hb * p.Wb * 32;
    SDN_REQUIRE(p.chunk_bytes < (1l << 32), "sdn_conv: frame too large (a channel chunk of a plane must stay below 4 GiB)");
    p.row_inc = (unsigned)(p.Wb - 2) * 32u;
    p.chunk_inc = (unsigned)(p.chunk_bytes - 2l * (p.Wb + 1) * 32);
    p.gx = sdn::div_up(W, PATCH_W);
    p.gy = sdn::div_up(H, PATCH_H);
    p.n_groups = p.gx * p.gy;
    int wg = n_workgroups > 0 ? n_workgroups : 256;
    if (wg > p.n_groups) wg = p.n_groups;
    // epilogue variant: the combinations the render CNN's plane-to-plane layers use are specialised
    const bool rp = resid_hi != nullptr, md = mod_w != nullptr, pj = proj_w != nullptr, lo = out_lo != nullptr;
    int epi = 255;
    if (!resid && !out_f32 && !pj && out_hi) {
        if (bias && !rp && !md) epi = lo ? 16 : 0;                 // conv1, conv2a, conv3a, conv4a
        else if (!bias && rp && md && lo) epi = 1 | 2 | 8 | 16;    // conv2b, conv3b (bias-free in the reference)
    }
    int dbg = 0;
#ifdef SDN_MLP_ABLATION
    if (const char *e1 = getenv("SDN_CONV_DBG")) dbg = atoi(e1);   // timing experiments only (re-read per call); results are wrong
#endif
    const dim3 grid(wg), block(64 * WAVES);
    hipStream_t hs = (hipStream_t)stream;
#define SDN_LAUNCH(TAPS, DBG, TERMS, EPI) hipLaunchKernelGGL((conv_kernel<TAPS, DBG, TERMS, EPI>), grid, block, 0, hs, p)
    if (taps == 1) {
        if (epi == 16) SDN_LAUNCH(1, 0, 3, 16);
        else SDN_LAUNCH(1, 0, 3, 255);   // (conv4b + projection)
    } else if (terms == 1) {
        if (dbg && (epi == 0 || epi == 16)) {
#ifdef SDN_MLP_ABLATION
            switch (dbg) {
                case 1: SDN_LAUNCH(9, 1, 1, 16); break;
                case 2: SDN_LAUNCH(9, 2, 1, 16); break;
                case 3: SDN_LAUNCH(9, 3, 1, 16); break;
                case 16: SDN_LAUNCH(9, 16, 1, 16); break;
                case 17: SDN_LAUNCH(9, 17, 1, 16); break;
                case 19: SDN_LAUNCH(9, 19, 1, 16); break;
                case 64: SDN_LAUNCH(9, 64, 1, 16); break;
                case 65: SDN_LAUNCH(9, 65, 1, 16); break;
                case 67: SDN_LAUNCH(9, 67, 1, 16); break;
                default: SDN_LAUNCH(9, 0, 1, 16); break;
            }
#endif
        } else if (epi == 0) SDN_LAUNCH(9, 0, 1, 0);
        else if (epi == 16) SDN_LAUNCH(9, 0, 1, 16);
        else if (epi == 27) SDN_LAUNCH(9, 0, 1, 27);
        else SDN_LAUNCH(9, 0, 1, 255);
    } else {
        if (epi == 16) SDN_LAUNCH(9, 0, 3, 16);
        else if (epi == 27) SDN_LAUNCH(9, 0, 3, 27);
        else SDN_LAUNCH(9, 0, 3, 255);
    }
#undef SDN_LAUNCH
    return sdn::check_launch("sdn_conv");
}

}  // extern "C"
