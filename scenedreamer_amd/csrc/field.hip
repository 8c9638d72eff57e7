// Fused field renderer for gfx950: sample placement -> hash-grid lookup -> style-modulated MLP on
// MFMA -> volume rendering + sky compositing.  Fast path for the reference's
// Generator._forward_perpix / _forward_perpix_sub (imaginaire/generators/scenedreamer.py:285-430)
// with mc_utils.sample_depth_batched (:82-151), GridEncoder.forward (gridencoder/grid.py:140-156),
// LightningMLP.forward (imaginaire/model_utils/layers.py:92-126) and volum_rendering_relu
// (mc_utils.py:154-161).
//
// Kernels
//   collapse_kernel   once per scene.  The two trailing hash-grid coordinates (global_enc) are
//                     constant per scene, and every level is hashed into a power-of-two table, so
//                     (h ^ K) & m == (h & m) ^ (K & m) and the 32-corner 5-D blend factorises exactly
//                     into an 8-corner 3-D blend of a per-scene table
//                        T'[l][i] = sum_{c3,c4} w3 w4 T[l][i ^ K(c3,c4)]
//                     (SURVEY.md appendix A).  4x fewer gathers per sample; only the fp32 summation
//                     order differs from the reference.
//   pack_kernel       once per style code.  Folded MLP weights (W * alpha) are split into f16 hi + f16 lo
//                     and laid out in MFMA A-fragment order, so a wave fetches one fragment as one
//                     fully coalesced 1 KiB access and no shuffles are needed anywhere.
//   pack_mx_kernel    the colour layers' part of that stream as f16 hi fragments + block-scaled fp6 fragments
//                     of Wlo and Whi (layer8x).
//   encode_kernel     per frame, one wave per 8 rays, 4 samples of every ray per step (32 MFMA columns):
//                     places the samples (bit-identical decisions to the reference), blends the 16
//                     levels from the collapsed table and writes the features as the MLP's B fragments,
//                     already split into f16 hi / lo.  Reads the frame-wide ray arrays through a ray window.
//                     Pure gather: bound by L2 / Infinity-Cache / HBM bandwidth.
//   mlp_kernel        per frame, persistent, one wave per SIMD (4 per CU), 32 samples per wave step.
//                     The MLP is evaluated TRANSPOSED: D^T[feature][sample] = W[feature][k] * X[k][sample],
//                     weights are the MFMA A operand, samples the B operand.  A wave keeps all 256
//                     activations of its 32 samples in registers; the C/D register layout of one layer is
//                     consumed directly as the B layout of the next (the k-permutation this implies is
//                     baked into the packed weights), so activations never touch LDS or HBM.
//                     Precision: the north star demands 1e-3 abs on radiance and the density head
//                     amplifies hidden-activation error by ~1e2, which plain f16/bf16 MFMA misses by 10x
//                     (measured).  Every product is therefore evaluated as a 3-term split
//                     (Whi*Xhi + Wlo*Xhi + Whi*Xlo, f32 accumulate): ~2^-21 relative error at 3 f16 MFMAs
//                     per tile, 5.3x the rate of the f32 MFMA.  In the colour layers fc_5 / fc_6, whose error
//                     nothing amplifies, the two correction terms run as block-scaled fp6 products
//                     (v_mfma_scale_f32_32x32x64_f8f6f4: K = 64 at the issue cost of a K = 16 f16 MFMA).
//                     Volume rendering, the density head, label bias, clamp and sky blend run in the
//                     epilogues on the VALU.  The persistent workgroups draw their 32-ray groups from a
//                     ticket counter; groups whose rays all miss are skipped.
//   field_kernel      = mlp_kernel<.., FUSED>: the north star's "hash-grid lookup plus the MLP fused into ONE kernel".  Every
//                     pass starts with the encode stage of its own 32 samples (sample placement + collapsed-table gathers,
//                     the very device functions encode_kernel is made of) executed by the MLP wave into its B-fragment
//                     registers -- a lane's 8 levels x 8 channels ARE its 8 B fragments, so nothing is exchanged -- while the
//                     accumulator / fragment-ring registers are dead.  No feature buffer, no 10.8 GB HBM round trip.
//   sky_kernel        the same machinery for SKYMLP on every ray of the padded frame + the frame mean.
//   head_kernel       the render CNN's first layer on the same machinery: net_out rows -> conv1 -> LeakyReLU -> activation planes.
//   chain_kernel      the render CNN's 1x1 tail (conv4a -> conv4b + residual -> conv4 -> tanh) as a register-resident per-pixel MLP.
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <utility>

#include "sdn_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

constexpr int HID = 256;        // hidden width (layers.py:62)
constexpr int FEAT = 128;       // hash-grid output width: 16 levels x 8 channels
constexpr int OUTC = 64;        // colour feature width (final_feat_dim)
constexpr int NLEV = 16;
constexpr int NLAB = 12;
constexpr int RAYS_PER_TILE = 8;
constexpr int SAMP_PER_STEP = 4;
constexpr int MAXM = 8;
constexpr int MAX_LIN = 80;     // up to 78 samples per ray
constexpr float ACT_SCALE = 0.4f; // LeakyReLU_0.2(x) = 0.4 * (1.5 x + |x|)
// The packed weights of the trunk layers fc_1 .. fc_4 carry 2^TRUNK_SHIFT (pack_kernel has the reason); the MLP kernel takes
// the factor back out of their accumulators in the activation's bias fma.  -DSDN_TRUNK_SHIFT=0 is the ablation build.
#ifndef SDN_TRUNK_SHIFT
#define SDN_TRUNK_SHIFT 8
#endif
constexpr int TRUNK_SHIFT = SDN_TRUNK_SHIFT;
constexpr float TRUNK_K = 1.0f / (float)(1 << TRUNK_SHIFT);

// ---- packed weight layout (in units of half8 = one lane's fragment) ---------------------------------
// layer 0: fc_1   K=128 -> 8 k-steps, 8 row blocks
// layer 1..5: fc_2..fc_6  K=256 -> 16 k-steps, 8 row blocks
// layer 6: fc_out_c  K=256 -> 16 k-steps, 2 row blocks
// fragment f of unit u of a layer sits at (u * 4 + f) * 64 + lane (unit order: see unit_coords)
constexpr size_t L0_FRAGS = 8 * 8 * 2 * 64;
constexpr size_t LH_FRAGS = 16 * 8 * 2 * 64;
constexpr size_t LO_FRAGS = 16 * 2 * 2 * 64;
constexpr size_t PACKED_FRAGS = L0_FRAGS + 5 * LH_FRAGS + LO_FRAGS;

// ---- fp32 constant block ----------------------------------------------------------------------------
constexpr int C_LABEL_BIAS = 0;                      // [12][256]  fc_m_a^T + fc_1.bias
constexpr int C_BETA = C_LABEL_BIAS + NLAB * HID;    // [5][256]   ModLinear output bias
constexpr int C_WSIGMA = C_BETA + 5 * HID;           // [256]
constexpr int C_BC = C_WSIGMA + HID;                 // [64]
constexpr int C_BSIGMA = C_BC + OUTC;                // [1]
constexpr int C_SKY_AVG = C_BSIGMA + 4;              // [64]
constexpr int C_TOTAL = C_SKY_AVG + OUTC;

// The rays of a launch are a WINDOW of the ray arrays the ray marcher wrote for the whole padded frame (the field is
// evaluated on the 4-px apron the image can depend on, a band of rows, or a chunk of either): local ray r is ray
// w = ray0 + r of a window of `cols` columns whose ray (y, x) is source ray first + y * pitch + x.  No window: cols = 0,
// source ray = ray0 + r.  Reading through the window replaces four strided-slice copies per frame on the host side.
//
// Ray ORDER of a launch (tiled_bx > 0): launch-local ray r is not pixel r of the window in row-major order but pixel
//   (4 * (b / tiled_bx) + (r & 31) / 8,  8 * (b % tiled_bx) + (r & 7)),  b = r / 32:
// the 32 rays a workgroup takes through a pass together (4 waves x 8 rays) are an 8 x 4 pixel BLOCK instead of 32 consecutive
// pixels of a row.  What a group can leave out -- the colour branch of a pass whose samples all have weight zero, the passes
// behind its last ray's saturation -- it leaves out when ALL its rays agree, and rays that are neighbours in both directions
// agree more often: measured on the benchmark frames 36 -> 40 %, 34 -> 40 %, 23 -> 32 % of the passes without colour branch
// (tools/dbg_sigma_stats.py).  A wave still reads 8 consecutive rays of a row (the same coalescing), rays are independent
// (net_out is the same bits per ray, only its per-group statistics move), the per-ray OUTPUTS stay in the window's row-major
// order (out_row).  The host sets it when the launch covers a whole window of 8k columns x 4m rows -- or (sdn_field_render only,
// `rows` > 0) a whole window of ANY size: the block grid is ceil(cols / 8) x ceil(rows / 4), the launch has that many 32-ray groups,
// and the block positions outside the window are no rays (`valid`): the reference's 570 x 990 padded frame (990 = 8 * 123 + 6)
// takes the blocked order that way.
struct RayWindow {
    int32_t n_src;             // rays in the source arrays (stride of depth2's two planes)
    int32_t pitch, first, cols, ray0;
    int32_t tiled_bx;          // 0: row-major ray order; else 8 x 4 pixel blocks, this many per block row (= ceil(cols / 8))
    int32_t rows;              // tiled + ragged: rows of the window (block positions at x >= cols or y >= rows are no rays); else 0
    // window-local pixel index (row-major) of launch-local ray r; -1: a block position outside a ragged window
    __device__ __forceinline__ int pix(int r) const {
        const int w = ray0 + r;
        if (tiled_bx == 0) return w;
        const int b = w >> 5, by = b / tiled_bx, bxi = b - by * tiled_bx;
        const int y = 4 * by + ((w & 31) >> 3), x = 8 * bxi + (w & 7);
        if (rows > 0 && (x >= cols || y >= rows)) return -1;
        return y * cols + x;
    }
    __device__ __forceinline__ bool valid(int r) const { return rows == 0 || pix(r) >= 0; }
    __device__ __forceinline__ int src(int r) const {
        int q = pix(r);
        q = q < 0 ? 0 : q;     // (a position outside the window reads the window's first ray: a valid address, discarded by the caller)
        return cols > 0 ? first + (q / cols) * pitch + (q % cols) : q;
    }
    // row of launch-local ray r in the launch's per-ray outputs / inputs that are NOT read through the window (net_out, the
    // per-sample outputs, the stratified randoms u): r itself, or -- tiled, where ray0 == 0 -- the pixel's row-major index
    __device__ __forceinline__ int out_row(int r) const { return tiled_bx == 0 ? r : pix(r); }
};


struct EncParams {
    const int32_t *voxel_id;   // [R, M]
    const float *depth2;       // [2, R, M]
    const float *raydirs;      // [R, 3]
    const uint8_t *lut;        // [1024] minecraft id -> reduced label (ignore already mapped to dirt)
    const float *table3;       // [16][T][8] collapsed table
    float *feat;               // [n_tiles][nch][8 k-steps][64 lanes][8 f16 hi | 8 f16 lo]: the MLP's B fragments, split
    float *dist;               // [n_tiles][nch][32]  new_dists * dists_scale (0 for padding samples)
    uint8_t *label;            // [n_tiles][nch][32]
    uint8_t *rayflag;          // [R] bit0 sky_only, bit1 nosky
    int32_t R, M, ns, nch, n_tiles;
    uint32_t tmask;            // T - 1
    float ori[3], delim[3];
    float sample_depth, dists_scale;
    int32_t genc_oob;          // global_enc outside [0,1] after mapping: every feature is zero
    int32_t ieee_div;          // stochastic sampling: rand / nsamples as an IEEE division (CPU reference) instead of * (1/n)
    const float *lin;          // dev [ns+1]  deterministic: linspace(0,1,ns+3)[1:-1]; stochastic: linspace(0,1,ns+2)[:-1]
    const float *u;            // dev [R][ns+1] uniform randoms of the training-time stratified sampling, or nullptr
    const float *scales;       // dev [16]    per-level scale, exp2f(l*S)*H-1 evaluated on the host
    RayWindow win;             // where ray r of this launch lives in voxel_id / depth2 / raydirs
};

struct MlpParams {
    const float *feat;
    const float *dist;
    const uint8_t *label;
    const uint8_t *rayflag;
    const half8 *wpk;          // packed weights
    const float *consts;       // fp32 constant block
    const float *sky_c;        // [R, 64] sky_net output per ray
    float *net_out;            // [R, 64]
    int32_t R, ns, nch, n_tiles;
    float term_depth;          // early ray termination: optical depth -ln(eps) beyond which a ray is opaque; <= 0: off
    uint8_t *passes;           // optional [ceil(n_tiles / 4)]: passes every 32-ray group went through (tests / bench)
    RayWindow win;             // sky_c is indexed with the SOURCE ray (it covers the whole padded frame)
    const float *sky_avg;      // optional dev [64]: frame mean of sky_c (else the value inside `consts`)
    int32_t *ticket;           // optional dev int32[2], zero before the first launch (the kernel leaves it zero): the
                               // persistent workgroups draw their 32-ray groups from it instead of taking every
                               // gridDim.x-th one (a static share that is mostly sky leaves its workgroup idle at the end)
    EncParams enc;             // FUSED (field_kernel): the encode stage's inputs; feat / dist / label / rayflag are unused then
    const float *cam_ori_dev;  // FUSED, optional dev f32 [3]: the camera origin read from device memory (overrides enc.ori): callers that hold
                               // it as a device tensor (Generator._forward_perpix's cam_ori_t) need no device -> host copy per call
    // MODE_FUSED_AUX: the other return values of Generator._forward_perpix (scenedreamer.py:429-430), each optional
    float *w_out;              // [R][ns]     weights: volume-rendering weight of every sample, * !sky_only (:373-376)
    float *depth_out;          // [R][ns]     rand_depth after the NaN / inf -> 0 replacement (:350-352)
    float *sig_out;            // [R][ns]     net_out_s: fc_sigma's output per sample (layers.py:114)
    float *col_out;            // [R][ns][64] net_out_c: fc_out_c's output per sample (layers.py:124)
    float *skyb_out;           // [R][64]     skynet_out_c after the keep_sky_out blend with sky_avg (:401)
    uint8_t *nosky_out;        // [R]         nosky_mask (:382-383)
    float *sigma_out;          // MODE_RAW: [R] density fc_sigma(f) of every row (LightningMLP.forward's first output)
    uint8_t *colour_passes;    // optional [ceil(n_tiles / 4)]: passes of every 32-ray group that ran the colour branch (tests / bench)
    int32_t no_colour_skip;    // field_kernel: 1 = evaluate fc_5 / fc_6 / fc_out_c in every pass (A/B switch; the results are identical)
};

// mlp_kernel's input / output modes
constexpr int MODE_BUFFER = 0;      // features from encode_kernel's buffer
constexpr int MODE_FUSED = 1;       // field_kernel: every pass encodes its own samples
constexpr int MODE_FUSED_AUX = 2;   // field_kernel that also writes the per-sample weights and depths (Generator._forward_perpix's
                                    // `weights` / `rand_depth` outputs, used by inference_givenstyle_depth, scenedreamer.py:812-817)
constexpr int MODE_RAW = 3;         // LightningMLP.forward as an op (imaginaire/model_utils/layers.py:92-126): rows of 128 f32 features
                                    // + a label per row in, (sigma, colour features) per row out; no sample placement, no compositing.
                                    // Row (tile, ch, j) = tile * 256 + ch * 32 + j: `R` counts rows, nch = 8, a tile = 256 rows.

// Exchanges inside a quad of lanes (the 4 samples of a ray in a pass) as DPP operands of the consuming VALU instruction:
// __shfl_* compiles to ds_bpermute_b32, an LDS round trip per exchange (64 of them in the volume-rendering epilogue of a pass).
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int quad_dpp(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
constexpr int QUAD_XOR1 = 0xB1;    // quad_perm [1,0,3,2]: lane ^ 1
constexpr int QUAD_XOR2 = 0x4E;    // quad_perm [2,3,0,1]: lane ^ 2
constexpr int QUAD_UP1 = 0x90;     // quad_perm [0,0,1,2]: lane - 1 (lane 0 of the quad reads itself)
constexpr int QUAD_UP2 = 0x44;     // quad_perm [0,1,0,1]: lane - 2 (lanes 0, 1 read themselves)
constexpr int QUAD_LAST = 0xFF;    // quad_perm [3,3,3,3]: the quad's last lane
constexpr int DPP_ROW_HALF_MIRROR = 0x141;   // lane i of a row of 16 reads lane i ^ 7 (reversal inside each group of 8)
constexpr int DPP_ROW_MIRROR = 0x140;        // lane i reads lane 15 - i

// =====================================================================================================
// collapse
// =====================================================================================================
struct CollapseParams {
    const float *emb;   // original table, level l at emb + off[l] * 8
    float *table3;
    uint32_t T;
    uint32_t off[NLEV];
    uint32_t K[NLEV][4];   // hash contribution of corner (c3, c4), index c3 + 2*c4, already & (T-1)
    float w[NLEV][4];      // w3 * w4 with the reference's multiply order
};

__global__ __launch_bounds__(256) void collapse_kernel(const CollapseParams p) {
    const uint32_t level = blockIdx.y;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.T) return;
    const float *src = p.emb + (size_t)p.off[level] * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) {  // c3 fastest, like the corner index of the reference loop
        const float4 a = *reinterpret_cast<const float4 *>(src + (size_t)(i ^ p.K[level][c]) * 8);
        const float4 b = *reinterpret_cast<const float4 *>(src + (size_t)(i ^ p.K[level][c]) * 8 + 4);
        const float w = p.w[level][c];
        acc[0] += w * a.x; acc[1] += w * a.y; acc[2] += w * a.z; acc[3] += w * a.w;
        acc[4] += w * b.x; acc[5] += w * b.y; acc[6] += w * b.z; acc[7] += w * b.w;
    }
    float *dst = p.table3 + ((size_t)level * p.T + i) * 8;
    *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4 *>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// =====================================================================================================
// weight packing
// =====================================================================================================
// k index that element e of lane-half h holds in k-step s of the B operand
__host__ __device__ inline int kmap_first(int s, int h, int e) { return 16 * s + 8 * h + e; }
__host__ __device__ inline int kmap_hidden(int s, int h, int e) {
    // C/D layout of v_mfma_f32_32x32x16: register r of lane-half h holds row (r&3) + 8*(r>>2) + 4*h;
    // k-step s consumes registers 8*(s&1) .. 8*(s&1)+7 of row block s>>1
    return 32 * (s >> 1) + 16 * (s & 1) + (e & 3) + 8 * (e >> 2) + 4 * h;
}

// Unit order of the packed stream.  A unit = the 4 fragments (ib,hi) (ib,lo) (ib+1,hi) (ib+1,lo) of one k-step
// for a pair of 32-row output blocks, 4 KiB; 4 consecutive units form one 16-KiB LDS ring slot.
//   8-row-block layers: the UPPER half of the outputs (row blocks 0-3) for all k-steps comes first, then the
//   lower half (4-7): unit u -> half = u / (2*NS), s = (u % (2*NS)) / 2, ib = 4*half + 2*(u & 1).
//   This order is what lets mlp_kernel hide every activation epilogue behind MFMAs (see there).
//   output layer (2 row blocks): unit u = k-step u.
__host__ __device__ inline void unit_coords(int nib, int ns, int u, int &s, int &ib0) {
    if (nib == 8) {
        const int half = u / (2 * ns), rem = u % (2 * ns);
        s = rem >> 1;
        ib0 = 4 * half + 2 * (rem & 1);
    } else {
        s = u;
        ib0 = 0;
    }
}

struct PackParams {
    const float *w1;      // [256,128]
    const float *wh[5];   // [256,256] each, W * alpha already folded
    const float *wc;      // [64,256]
    half8 *out;
};

__global__ __launch_bounds__(256) void pack_kernel(const PackParams p) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (layer, unit, row block of the pair, lane)
    const size_t n0 = 32 * 2 * 64, nh = 64 * 2 * 64, no = 16 * 2 * 64;
    if (g >= n0 + 5 * nh + no) return;
    int layer, nib, ns, K;
    const float *W;
    size_t base, r = g;
    if (r < n0) {
        layer = 0; nib = 8; ns = 8; K = FEAT; W = p.w1; base = 0;
    } else if (r < n0 + 5 * nh) {
        r -= n0; layer = 1 + (int)(r / nh); r %= nh; nib = 8; ns = 16; K = HID; W = p.wh[layer - 1];
        base = L0_FRAGS + (size_t)(layer - 1) * LH_FRAGS;
    } else {
        r -= n0 + 5 * nh; layer = 6; nib = 2; ns = 16; K = HID; W = p.wc; base = L0_FRAGS + 5 * LH_FRAGS;
    }
    const int lane = (int)(r % 64); r /= 64;
    const int sel = (int)(r % 2);
    const int u = (int)(r / 2);
    int s, ib0;
    unit_coords(nib, ns, u, s, ib0);
    const int row = 32 * (ib0 + sel) + (lane & 31), h = lane >> 5;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int k = layer == 0 ? kmap_first(s, h, e) : kmap_hidden(s, h, e);
        // layers 1..6 consume a' = 1.5 x + |x| = LeakyReLU_0.2(x) / 0.4 (one v_fma instead of mul + max in the
        // MLP kernel's activation), so their weights carry the factor 0.4
        // The trunk layers (fc_1 .. fc_4, whose error the density head amplifies) are stored times 2^TRUNK_SHIFT: the lo
        // part of a weight of magnitude 0.03 is ~7e-6, deep in f16's subnormal range (quantum 6e-8), which left the split
        // weight with ~20 significant bits instead of 22; scaled by 2^shift the quantum shrinks by as much.  The kernel
        // takes the factor back out in the bias fma of the activation (act_stage, stage 1) -- no extra instruction.
        const float v = W[(size_t)row * K + k] * (layer == 0 ? 1.0f : ACT_SCALE) * (layer <= 3 ? (float)(1 << TRUNK_SHIFT) : 1.0f);
        const _Float16 vh = (_Float16)v;
        hi[e] = vh;
        lo[e] = (_Float16)(v - (float)vh);
    }
    p.out[base + ((size_t)u * 4 + 2 * sel + 0) * 64 + lane] = hi;
    p.out[base + ((size_t)u * 4 + 2 * sel + 1) * 64 + lane] = lo;
}

// ---- MX variant of the packed stream: fc_5 / fc_6 (packed layers 4 and 5) in the layout of layer8x ---------------------
// fp6 e2m3 code of |v| <= 7.5 (round to nearest even; the 32 non-negative codes are contiguous in value order)
__device__ inline unsigned fp6_code(float v) {
    const float a = fminf(fabsf(v), 7.5f);
    float c;
    if (a < 2.f) c = rintf(a * 8.f);                 // 0 .. 16: subnormals and the binade [1, 2), step 1/8
    else if (a < 4.f) c = 16.f + rintf((a - 2.f) * 4.f);
    else c = 24.f + rintf((a - 4.f) * 2.f);
    const unsigned code = (unsigned)fminf(c, 31.f);
    return code | (v < 0.f ? 32u : 0u);
}

struct PackMxParams {
    const float *wh[4];   // up to 4 hidden layers' weights [256,256] (field: fc_5, fc_6 with alpha folded; sky: fc2..fc5)
    int n_layers;
    size_t base;          // fragment index of the first of those layers in the packed stream
    half8 *out;           // the packed stream (all layers already written by pack_kernel / sky_pack_kernel)
};

__global__ __launch_bounds__(256) void pack_mx_kernel(const PackMxParams p) {
#pragma clang fp contract(off)   // hi = f16(f32(W * 0.4)) in both branches: a fused multiply would break exact ties differently
    const int g = blockIdx.x * 256 + threadIdx.x;    // one thread per (layer, unit, lane)
    if (g >= p.n_layers * 64 * 64) return;
    const int lane = g % 64, u = (g / 64) % 64, layer = g / (64 * 64);
    const float *W = p.wh[layer];
    half8 *out = p.out + p.base + (size_t)layer * LH_FRAGS + (size_t)u * 4 * 64;
    const int half = u / 32, kb = (u % 32) / 8, sub = u % 8, ib0 = 4 * half, h = lane >> 5;
    if (sub < 4) {   // f16 hi fragments of k-step 4 kb + sub for the half's 4 row blocks
        const int s = 4 * kb + sub;
        for (int f = 0; f < 4; f++) {
            const int row = 32 * (ib0 + f) + (lane & 31);
            half8 hi;
            for (int e = 0; e < 8; e++) hi[e] = (_Float16)(W[(size_t)row * HID + kmap_hidden(s, h, e)] * ACT_SCALE);
            out[f * 64 + lane] = hi;
        }
        return;
    }
    const int term = (sub - 4) / 2, iba = ib0 + 2 * ((sub - 4) % 2);   // term 0: Wlo (x x6), term 1: Whi (x xl6)
    for (int rb = 0; rb < 2; rb++) {
        const int row = 32 * (iba + rb) + (lane & 31);
        float v[32], vmax = 0.f;
        for (int i = 0; i < 32; i++) {
            const float w = W[(size_t)row * HID + kmap_hidden(4 * kb + i / 8, h, i % 8)] * ACT_SCALE;
            const float hi = (float)(_Float16)w;
            v[i] = term == 0 ? w - hi : hi;
            vmax = fmaxf(vmax, fabsf(v[i]));
        }
        int e = 0;   // smallest power of two with vmax <= 7.5 * 2^e
        if (vmax > 0.f) {
            e = (int)floorf(log2f(vmax / 7.5f)) - 1;
            while (ldexpf(7.5f, e) < vmax) e++;
        }
        if (e < -126) e = -126;
        unsigned w6[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        for (int i = 0; i < 32; i++) {
            const unsigned long long code = fp6_code(ldexpf(v[i], -e));
            const int bit = 6 * i, d = bit >> 5, o = bit & 31;
            w6[d] |= (unsigned)(code << o);
            if (o > 26) w6[d + 1] |= (unsigned)(code >> (32 - o));
        }
        u32x4v f0 = {w6[0], w6[1], w6[2], w6[3]}, f1 = {w6[4], w6[5], (unsigned)(127 + e), 0u};
        out[(2 * rb) * 64 + lane] = __builtin_bit_cast(half8, f0);
        out[(2 * rb + 1) * 64 + lane] = __builtin_bit_cast(half8, f1);
    }
}

// f32 -> (hi, lo) f16 pair with hi + lo == x to ~2^-22 relative.  hi is rounded to NEAREST (v_cvt_pk_f16_f32, two
// values per instruction, new in gfx950): for the full 3-term product the rounding mode of hi is irrelevant (lo
// absorbs the remainder), but a layer evaluated WITHOUT the Whi.Xlo term (TERMS == 2 below) sees |x - hi| as its error:
// half as large and unbiased with round-to-nearest (tools/precision_study.py: 2.2x less output error than with
// v_cvt_pkrtz).  x - float(hi) is a single v_fma_mix_f32 reading the f16 half directly.
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ fp16x2 cvt_rtn(float a, float b) {
    return __builtin_bit_cast(fp16x2, __builtin_convertvector(float2v{a, b}, half2v));
}

__device__ __forceinline__ void split8(const float (&v)[8], half8 &hi, half8 &lo) {
    unsigned int hw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const fp16x2 hp = cvt_rtn(v[e], v[e + 1]);
        const fp16x2 lp = cvt_rtn(v[e] - (float)hp[0], v[e + 1] - (float)hp[1]);
        hw[e / 2] = __builtin_bit_cast(unsigned int, hp);
        lw[e / 2] = __builtin_bit_cast(unsigned int, lp);
    }
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h4 = {hw[0], hw[1], hw[2], hw[3]}, l4 = {lw[0], lw[1], lw[2], lw[3]};
    hi = __builtin_bit_cast(half8, h4);
    lo = __builtin_bit_cast(half8, l4);
}

// =====================================================================================================
// encode: sample placement + collapsed hash-grid lookup
// =====================================================================================================
struct RayBoxes {
    float t[MAXM], t2[MAXM];
    int32_t id[MAXM];
};

// Everything that feeds a DISCRETE decision of the reference (box index of a sample, the is_gnd test,
// the grid cell) is evaluated with exactly the reference's fp32 operation sequence: no FMA contraction.
struct Placed {
    float depth, dist;
    int idx;
    float raw_depth;   // heads + midpoints before the NaN / inf -> 0 replacement
    int raw_idx;       // #{accu < mid} before clamping to M - 1
};

// Position of stratified point i in [0,1) (mc_utils.py:116-125): deterministic -> lin[i]; stochastic (training)
// -> rand / nsamples + linspace(0, 1, nsamples + 1)[i], `u` being the caller's torch.rand draw for this ray.
// `rand_samples / nsamples` (tensor / Python scalar) is evaluated by PyTorch as a MULTIPLICATION by the float32 reciprocal on
// a CUDA tensor (BinaryDivTrueKernel.cu: CPU-scalar fast path, a * (1 / b)) and as an IEEE division on a CPU tensor; the
// two differ by 1 ulp when nsamples is not a power of two.  n_div = +nsamples: the CUDA reference's form (strat_division
// 0 of the entry points, the default of the host wrappers); n_div = -nsamples: the CPU form (strat_division 1), which the
// goldens recorded from the reference's CPU run were produced with.
__device__ __forceinline__ float strat_pos(const float *lin, const float *u, int n_div, int i) {
#pragma clang fp contract(off)
    if (u == nullptr) return lin[i];
    const float q = n_div > 0 ? u[i] * (1.0f / (float)n_div) : u[i] / (float)(-n_div);
    return q + lin[i];
}

__device__ __forceinline__ Placed place_sample(const RayBoxes &rb, int M, const float *lin, const float *u, int n_div, int sidx,
                                               float sample_depth) {
#pragma clang fp contract(off)
    // mc_utils.py:101-107.  torch.cumsum on the CPU (what the oracle and the golden vectors were produced
    // with) accumulates float32 inputs in double and rounds every prefix back to float; it is mirrored here
    // because a 1-ulp change of a sample depth moves a fine-level feature by up to ~1e-4.
    float accu[MAXM];
    double run_d = 0.0;
    float run = 0.f;
#pragma unroll
    for (int k = 0; k < MAXM; k++) {
        if (k < M) {
            float d = rb.t2[k] - rb.t[k];
            if (d != d) d = 0.f;
            run_d += (double)d;
            run = (float)run_d;
            accu[k] = run;
        } else {
            accu[k] = 0.f;
        }
    }
    const float total = fminf(run, sample_depth);
    // :118-135 deterministic stratified points and their midpoints
    const float s0 = strat_pos(lin, u, n_div, sidx) * total, s1 = strat_pos(lin, u, n_div, sidx + 1) * total;
    const float mid = (s1 + s0) / 2.f;
    Placed o;
    o.dist = s1 - s0;
    int idx = 0;
#pragma unroll
    for (int k = 0; k < MAXM; k++)
        if (k < M && mid > accu[k]) idx++;  // :139
    // :142-145 head of the box the sample falls into: t[0] + cumulative gaps
    float head = rb.t[0];
    double cg_d = 0.0;
#pragma unroll
    for (int k = 1; k < MAXM; k++) {
        if (k < M) {
            const float g = rb.t[k] - rb.t2[k - 1];
            cg_d += (double)g;
            const float cg = (float)cg_d;
            if (k == idx) head = cg + rb.t[0];
        }
    }
    float depth = head + mid;  // :149
    o.raw_depth = depth;
    o.raw_idx = idx;
    if (depth != depth || __builtin_isinf(depth)) depth = 0.f;  // scenedreamer.py:350-352
    o.depth = depth;
    o.idx = idx < M ? idx : M - 1;
    return o;
}

__device__ __forceinline__ float mul_add_exact(float a, float b, float c) {
#pragma clang fp contract(off)
    const float p = a * b;
    return p + c;
}

__device__ __forceinline__ float normalise_coord(float wc, float delim) {
#pragma clang fp contract(off)
    // scenedreamer.py:300 then grid.py:144:  ((wc / delim * 2 - 1) + 1) / 2
    float n = wc / delim;
    n = n * 2.f;
    n = n - 1.f;
    n = n + 1.f;
    return n / 2.f;
}

// ---- the per-sample steps of the encode stage, shared by encode_kernel (features handed to mlp_kernel through HBM) and by
// ---- field_kernel = mlp_kernel<.., FUSED> (the same steps at the start of every pass: lookup + MLP in ONE kernel) ------------
// Lane (h = lane >> 5, j = lane & 31) of a wave that owns ray tile `tile` works on ray tile * 8 + (j >> 2), sample
// 4 * ch + (j & 3) of pass ch, and on the levels 2 * s + h, s = 0..7: its 8 x 8 blended values ARE B fragment s of the MLP.
__device__ __forceinline__ void enc_load_ray(const EncParams &p, int rr, RayBoxes &rb, float (&d)[3]) {
    const size_t RS = (size_t)p.win.n_src;
#pragma unroll
    for (int k = 0; k < MAXM; k++) {
        if (k < p.M) {
            rb.t[k] = p.depth2[(size_t)rr * p.M + k];
            rb.t2[k] = p.depth2[(RS + rr) * p.M + k];
            rb.id[k] = p.voxel_id[(size_t)rr * p.M + k];
        } else {
            rb.t[k] = rb.t2[k] = __builtin_nanf("");
            rb.id[k] = 0;
        }
    }
    d[0] = p.raydirs[(size_t)rr * 3]; d[1] = p.raydirs[(size_t)rr * 3 + 1]; d[2] = p.raydirs[(size_t)rr * 3 + 2];
}

struct EncSample {
    float x0, x1, x2;     // grid coordinates in [0, 1]
    bool oob, valid, gnd; // outside the grid / a real sample of a real ray / world x <= 1 (scenedreamer.py:380)
    float dist;           // new_dists * dists_scale (0 for padding samples)
    int label;            // reduced label of the box the sample falls into
    float depth;          // rand_depth after the NaN / inf -> 0 replacement (only the AUX field kernel reads it)
};

__device__ __forceinline__ EncSample enc_place(const EncParams &p, const RayBoxes &rb, const float (&d)[3], int rl, int sidx, bool ray_ok) {
    EncSample e;
    e.valid = ray_ok && sidx < p.ns;
    const Placed pl = place_sample(rb, p.M, p.lin, p.u ? p.u + (size_t)p.win.out_row(rl) * (p.ns + 1) : nullptr, p.ieee_div ? -(p.ns + 1) : p.ns + 1,
                                   e.valid ? sidx : 0, p.sample_depth);
    const float wx = mul_add_exact(d[0], pl.depth, p.ori[0]);  // scenedreamer.py:354
    const float wy = mul_add_exact(d[1], pl.depth, p.ori[1]);
    const float wz = mul_add_exact(d[2], pl.depth, p.ori[2]);
    e.gnd = e.valid && wx <= 1.0f;                           // :380
    e.x0 = normalise_coord(wx, p.delim[0]);
    e.x1 = normalise_coord(wy, p.delim[1]);
    e.x2 = normalise_coord(wz, p.delim[2]);
    e.oob = p.genc_oob || e.x0 < 0.f || e.x0 > 1.f || e.x1 < 0.f || e.x1 > 1.f || e.x2 < 0.f || e.x2 > 1.f;
    e.dist = e.valid ? pl.dist * p.dists_scale : 0.f;
    e.depth = pl.depth;
    int id = rb.id[0];
#pragma unroll
    for (int k = 1; k < MAXM; k++) {
        id = k == pl.idx ? rb.id[k] : id;
        asm("" : "+v"(id));     // a select chain, not rb.id[pl.idx]: hipcc otherwise turns it into a dynamically indexed load
    }                           // and moves the ray's boxes to scratch memory (8 stores + 1 load per sample)
    e.label = p.lut[id & 1023];
    return e;
}

// The 8 blended channels of one level = 8 corners of the collapsed 3-D table, in three steps so that a caller can put the
// gathers of SEVERAL levels in flight before it blends any of them (field_kernel: one wave per SIMD has no other wave to hide
// a level's round trip behind): where the rows are, the rows, the blend in the reference's multiply order.
struct LevelAddr {
    float f0, f1, f2;       // fractional position inside the cell
    uint32_t off[8];        // byte offset of corner c's row from table3 (corner c: bit d of c = +1 on dimension d); 32 bits: the
                            // 16 x T x 32-byte table is far below 4 GiB, and a uniform base + 32-bit lane offset is the saddr form
                            // of global_load -- half the address registers of 64-bit pointers (64 instead of 128 per 4 levels)
};

__device__ __forceinline__ void enc_level_addr(const EncParams &p, const EncSample &e, int level, bool ok, LevelAddr &a) {
    const float scale = p.scales[level];
    float f0 = mul_add_exact(e.x0, scale, 0.5f), f1 = mul_add_exact(e.x1, scale, 0.5f), f2 = mul_add_exact(e.x2, scale, 0.5f);
    const float g0 = floorf(f0), g1 = floorf(f1), g2 = floorf(f2);
    a.f0 = f0 - g0; a.f1 = f1 - g1; a.f2 = f2 - g2;
    const uint32_t a0 = (uint32_t)g0, a1 = (uint32_t)g1 * 2654435761u, a2 = (uint32_t)g2 * 805459861u;
    const uint32_t b0 = a0 + 1u, b1 = a1 + 2654435761u, b2 = a2 + 805459861u;
    const uint32_t tb = (uint32_t)level * (p.tmask + 1u);     // first row of the level
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const uint32_t hsh = ((c & 1) ? b0 : a0) ^ ((c & 2) ? b1 : a1) ^ ((c & 4) ? b2 : a2);
        a.off[c] = (tb + (ok ? (hsh & p.tmask) : 0u)) * 32u;      // (lanes without a sample read row 0 and discard it)
    }
}

__device__ __forceinline__ void enc_level_load(const EncParams &p, const LevelAddr &a, float4 (&va)[8], float4 (&vb)[8]) {
    const char *base = reinterpret_cast<const char *>(p.table3);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        va[c] = *reinterpret_cast<const float4 *>(base + a.off[c]);
        vb[c] = *reinterpret_cast<const float4 *>(base + a.off[c] + 16);
    }
}

__device__ __forceinline__ void enc_level_blend(const LevelAddr &a, const float4 (&va)[8], const float4 (&vb)[8], float (&res)[8]) {
    // res[ch] = fma(w_c, row_c[ch], res[ch]) over the corners in order, two channels per v_pk_fma_f32 (the same fused
    // multiply-add per channel as the scalar form -- bit-identical -- in half the issue slots: this stage runs with one wave
    // per SIMD inside field_kernel, where every instruction is ~4 cycles of an idle matrix pipe)
    float2v r01 = {0.f, 0.f}, r23 = {0.f, 0.f}, r45 = {0.f, 0.f}, r67 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; c++) {
        float w = 1.f;  // same multiply order as gridencoder.cu:152-160
        w *= (c & 1) ? a.f0 : 1.f - a.f0;
        w *= (c & 2) ? a.f1 : 1.f - a.f1;
        w *= (c & 4) ? a.f2 : 1.f - a.f2;
        const float2v w2 = {w, w};
        r01 = __builtin_elementwise_fma(w2, float2v{va[c].x, va[c].y}, r01);
        r23 = __builtin_elementwise_fma(w2, float2v{va[c].z, va[c].w}, r23);
        r45 = __builtin_elementwise_fma(w2, float2v{vb[c].x, vb[c].y}, r45);
        r67 = __builtin_elementwise_fma(w2, float2v{vb[c].z, vb[c].w}, r67);
    }
    res[0] = r01[0]; res[1] = r01[1]; res[2] = r23[0]; res[3] = r23[1];
    res[4] = r45[0]; res[5] = r45[1]; res[6] = r67[0]; res[7] = r67[1];
}

// one level at a time (encode_kernel: its other waves hide the round trip); no loads at all for lanes without a sample
__device__ __forceinline__ void enc_level(const EncParams &p, const EncSample &e, int level, bool use_feat, float (&res)[8]) {
#pragma unroll
    for (int c = 0; c < 8; c++) res[c] = 0.f;
    if (!e.oob && e.valid && use_feat) {
        LevelAddr a;
        float4 va[8], vb[8];
        enc_level_addr(p, e, level, true, a);
        enc_level_load(p, a, va, vb);
        enc_level_blend(a, va, vb, res);
    }
}

// field_kernel's form: NB levels 2 * (S0 + t) + h, t = 0 .. NB-1, with all their gathers issued before the first blend, and
// branch-free: lanes without a sample gather row 0 and get zeros by selection.  The arithmetic of a lane that has a sample is
// enc_level's, bit for bit.  (NB = 4: 64 x 16 B in flight per lane, 2 round trips per pass.  A 2-deep software pipeline of
// 2-level batches was tried: its register pressure made hipcc spill, and a scratch reload is a vector-memory operation -- the
// wait for it drains every gather issued before it.  The stage moves 512 KiB per CU and pass from L2: ~4 us at the L2's
// 135 GB/s per CU whatever the schedule.)
template <int NB>
__device__ __forceinline__ void enc_levels(const EncParams &p, const EncSample &e, int s0, int h, bool use_feat, float (&res)[NB][8]) {
    const bool ok = !e.oob && e.valid && use_feat;
    LevelAddr a[NB];
    float4 va[NB][8], vb[NB][8];
#pragma unroll
    for (int t = 0; t < NB; t++) enc_level_addr(p, e, 2 * (s0 + t) + h, ok, a[t]);
#pragma unroll
    for (int t = 0; t < NB; t++) enc_level_load(p, a[t], va[t], vb[t]);
#pragma unroll
    for (int t = 0; t < NB; t++) {
        enc_level_blend(a[t], va[t], vb[t], res[t]);
#pragma unroll
        for (int c = 0; c < 8; c++) res[t][c] = ok ? res[t][c] : 0.f;
    }
}

#ifndef SDN_ENC_OCC
#define SDN_ENC_OCC 3   // 3 waves per SIMD = up to 168 VGPRs: no spills (with 1, hipcc aims at 4 waves and spills 100 B; 4 and 5 measured slower)
#endif
__global__ __launch_bounds__(256, SDN_ENC_OCC) void encode_kernel(const EncParams p) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= p.n_tiles) return;
    const int h = lane >> 5, j = lane & 31;
    const int ray = tile * RAYS_PER_TILE + (j >> 2);
    const bool ray_ok = ray < p.R && p.win.valid(ray);
    const int rl = ray_ok ? ray : p.R - 1;      // local ray (index into u / rayflag)
    const int rr = p.win.src(rl);               // the same ray in the source arrays

    RayBoxes rb;
    float d[3];
    enc_load_ray(p, rr, rb, d);
    bool gnd = false;
    // A ray that hits nothing gets weight 0 for all its samples (scenedreamer.py:376: weights * (1 - sky_only)), so
    // its features are never used: no gathers for its lanes, and no feature traffic at all for a tile of 8 such rays
    // (the MLP kernel reads whatever is there and discards the result by selection, not multiplication).
    const bool use_feat = ray_ok && rb.id[0] != 0;
    const bool tile_dead = !__any(use_feat);

    for (int ch = 0; ch < p.nch; ch++) {
        const EncSample e = enc_place(p, rb, d, rl, ch * SAMP_PER_STEP + (j & 3), ray_ok);
        if (e.gnd) gnd = true;
        const size_t tc = (size_t)tile * p.nch + ch;
        if (h == 0) {
            p.dist[tc * 32 + j] = e.dist;
            p.label[tc * 32 + j] = (uint8_t)e.label;
        }
        if (tile_dead) continue;
        float *fout = p.feat + (tc * 8 * 64 + lane) * 8;
#pragma unroll 2
        for (int s = 0; s < 8; s++) {
            float res[8];
            enc_level(p, e, 2 * s + h, use_feat, res);
            float *o = fout + (size_t)s * 64 * 8;
            // The features leave as the MLP's operands: the 8 values of this lane's B fragment split into f16 hi (first
            // 16 bytes) and f16 lo (second 16 bytes) -- the same 32 bytes per lane and k-step as 8 floats, and exactly the
            // split mlp_kernel used to do at every pass start (128 VALU instructions of a wave that has no issue slots to
            // spare; this kernel waits for gathers anyway).
            // Streaming stores: the 4.9 GB of features are written once and read once by the MLP kernel; they should not
            // displace the collapsed table (the gathers' working set) from L2 / Infinity Cache
            half8 hi8, lo8;
            split8(res, hi8, lo8);
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(__builtin_bit_cast(u4v, hi8), reinterpret_cast<u4v *>(o));
#ifndef SDN_ENC_HI_ONLY   // (timing ablation: upper bound of what fewer feature bytes can buy)
            __builtin_nontemporal_store(__builtin_bit_cast(u4v, lo8), reinterpret_cast<u4v *>(o + 4));
#endif
        }
    }
    // per-ray flags: any over the ray's 4 lanes
    gnd = quad_dpp<QUAD_XOR1>((int)gnd) | (int)gnd;
    gnd = quad_dpp<QUAD_XOR2>((int)gnd) | (int)gnd;
    if (h == 0 && (j & 3) == 0 && ray_ok) {
        const bool sky_only = rb.id[0] == 0;             // scenedreamer.py:337
        int last = rb.id[0];
#pragma unroll
        for (int k = 1; k < MAXM; k++)
            if (k == p.M - 1) last = rb.id[k];
        const bool nosky = (last != 0) || gnd;           // :335, :382
        p.rayflag[ray] = (uint8_t)((sky_only ? 1 : 0) | (nosky ? 2 : 0));
    }
}

// =====================================================================================================
// mc_utils.sample_depth_batched as an op of its own (the un-fused path and training): one thread per ray
// =====================================================================================================
struct SampleParams {
    const float *depth2;   // [2, R, M]
    const float *lin;      // [n_points]
    const float *u;        // [R, n_points] or nullptr
    float *rand_depth;     // [R, n_points - 1]
    float *new_dists;      // [R, n_points - 1]
    int64_t *idx;          // [R, n_points - 1]
    int32_t R, M, n_points;
    int32_t ieee_div;
    float sample_depth;
};

__global__ __launch_bounds__(256) void sample_depth_kernel(const SampleParams p) {
    const int ray = blockIdx.x * 256 + threadIdx.x;
    if (ray >= p.R) return;
    RayBoxes rb;
#pragma unroll
    for (int k = 0; k < MAXM; k++) {
        if (k < p.M) {
            rb.t[k] = p.depth2[(size_t)ray * p.M + k];
            rb.t2[k] = p.depth2[((size_t)p.R + ray) * p.M + k];
        } else {
            rb.t[k] = rb.t2[k] = __builtin_nanf("");
        }
        rb.id[k] = 0;
    }
    const float *u = p.u ? p.u + (size_t)ray * p.n_points : nullptr;
    const int ns = p.n_points - 1;
    for (int i = 0; i < ns; i++) {
        Placed pl = place_sample(rb, p.M, p.lin, u, p.ieee_div ? -p.n_points : p.n_points, i, p.sample_depth);
        // the op returns the RAW values: NaN depths of rays that hit nothing are zeroed by the caller (scenedreamer.py:350-352)
        // and the box index is the count itself, mc_utils.py:139 (place_sample clamps it for the label lookup)
        p.new_dists[(size_t)ray * ns + i] = pl.dist;
        p.rand_depth[(size_t)ray * ns + i] = pl.raw_depth;
        p.idx[(size_t)ray * ns + i] = pl.raw_idx;
    }
}

// =====================================================================================================
// MLP + compositing
// =====================================================================================================
// Structure of one pass (32 samples per wave through the 7 layers):
//
//  * weights: the 4 waves of a workgroup share ONE copy of the packed weight stream (1.47 MB per pass, identical
//    for every pass).  It flows L2 -> LDS by LDS-DMA (global_load_lds, 16 B/lane, no VGPRs) into a ring of
//    4 slots x 32 KiB (= 8 units = 48 MFMAs per wave); every wave issues 8 of a slot's 32 1-KiB pieces, 3 slots
//    ahead of use, one piece behind each of the first MFMAs after the slot's barrier.
//    Per slot: counted s_waitcnt vmcnt(8) (this wave's pieces of slots g and g+1 have landed) -> raw s_barrier
//    (everybody's have, and everybody is done with slot g-1) -> DMA for slot g+3 into the position of slot g-1.
//    (8 x 16 KiB slots, 7 ahead, measured 0.9 % slower on the same box: twice the barriers.)
//    Fragments go LDS -> registers by ds_read_b128 (lane-linear image: conflict-free) through a 3-unit register
//    ring that runs across slot boundaries.  LDS read traffic 85 B/clk/CU of 256, L2 -> LDS 21 B/clk/CU.
//    (A first version fetched fragments per wave from L2: 85 B/clk/CU through a 64 B/clk/CU path, 40.8 % MFMA busy.)
//  * the kernel runs ONE wave per SIMD (the 32 samples x 256 activations as hi+lo f16 and the 32 x 256 f32
//    accumulators take 256 of the 512 registers), so a wave gets one issue slot every ~4 cycles and anything that
//    is not interleaved with MFMAs is lost matrix time (measured: MFMA-only 10.4 ms + everything-else 12.1 ms =
//    22.5 ms when the activation epilogues ran between the layers).  Therefore the layer is evaluated as
//         upper half of the outputs (row blocks 0-3) for all k, then the lower half (4-7),
//    and the bias + LeakyReLU + f16 hi/lo re-split of a finished half is executed in the shadow of the MFMAs that
//    follow it: the lower half of layer l while layer l+1 starts on k-steps 0-7 (which only need the upper half),
//    the upper half of layer l+1 during its own last k-steps 8-15 of the lower half (k-steps 0-7 of its input are
//    dead by then, so the new B fragments overwrite them).  No second accumulator set is needed.
//  * density head, volume rendering, clamp, sky blend: VALU epilogue per pass / per ray tile.
constexpr int UNITS_PER_SLOT = 8;                  // one barrier per 8 units (48 MFMAs per wave)
constexpr int NSLOT = 4;
constexpr int SLOT_BYTES = UNITS_PER_SLOT * 4096;  // 32 KiB
constexpr int DMA_AHEAD = 3;
constexpr int PIECES = SLOT_BYTES / 4096;          // 1-KiB DMA pieces per wave and slot (4 waves)
constexpr int SLOTS_PER_PASS = (8 + 5 * 16 + 4) * 4 / UNITS_PER_SLOT;   // 46
constexpr int LDS_RING = 0;
constexpr int LDS_CONST = NSLOT * SLOT_BYTES;     // fp32 constant block
constexpr int LDS_FLAGS = LDS_CONST + ((C_TOTAL * 4 + 255) / 256) * 256;
// field_kernel only: the encode stage's small tables, so that a pass's sample placement waits for no dependent global load
constexpr int LDS_ENC_SCALES = LDS_FLAGS + 64;              // f32 [16]   per-level scales
constexpr int LDS_ENC_LIN = LDS_ENC_SCALES + NLEV * 4;      // f32 [MAX_LIN] stratified positions
constexpr int LDS_ENC_LUT = LDS_ENC_LIN + MAX_LIN * 4;      // u8 [1024]  block id -> reduced label
constexpr int LDS_TIMERS = LDS_ENC_LUT + 1024;            // u32 [16]  DBG & 512: cycles per segment of a pass (timing experiments)
constexpr int LDS_TOTAL = LDS_TIMERS + 64;

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const char glb_char;

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.f;
    return z;
}

struct Ring {
    int slots_per_pass;   // 92 for the field MLP, 72 for the sky MLP
    const char *wbytes;   // packed weights
    int g;                // slots consumed so far (uniform across the workgroup)
    int next_in_pass;     // slot-in-pass index of slot g + DMA_AHEAD
    int wave;             // wave index as a scalar (readfirstlane)
    int lane;
    int voff;             // per-lane byte offset of this wave's first piece inside a slot: wave*4096 + lane*16
    unsigned lds_lane;    // LDS byte address of this lane's 16 B inside fragment 0 of ring position 0
    int src_delta;        // (scalar) voff - lds_lane: what turns lds_lane into this lane's byte offset inside a slot
    int pend_global, pend_in_pass;   // slot whose refill was granted by the last ring_acquire (issued piecewise after it)
    const char *pend_src;            // this lane's source address of that refill's first piece
};

// Per-lane source address of a slot's first DMA piece = uniform slot base + this lane's offset (wave * 8 KiB + lane * 16).
// It is built HERE, per slot, by three VALU instructions inside one opaque asm block from the SGPR base and the lane's
// LDS address (a register every unit needs anyway), instead of leaving the arithmetic to hipcc: hipcc re-associates it
// into a kernel-long per-lane 64-bit base (wbytes + lane offset) plus a uniform slot offset -- two VGPRs for the whole
// kernel, which the fp6 variant spilled to scratch and reloaded at every refill behind `s_waitcnt vmcnt(0)`: a full drain
// of the DMA ring eleven times per pass in fc_5 alone (SQ_WAIT_ANY 11.5 % -> 17.6 % of the wave time).
__device__ __forceinline__ const char *ring_lane_src(const char *slot_base, const Ring &r) {
    unsigned int lo, hi;
    const unsigned int b_lo = (unsigned int)(size_t)slot_base, b_hi = (unsigned int)((size_t)slot_base >> 32);
    // lane offset = (lds_lane - lds_lane_base) + wave * PIECES * 1024, lds_lane_base + ... folded into `delta` (uniform)
    asm volatile("v_add_u32 %0, %2, %3\n\t"
                 "v_add_co_u32 %0, vcc, %4, %0\n\t"
                 "v_mov_b32 %1, %5\n\t"
                 "v_addc_co_u32 %1, vcc, 0, %1, vcc"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(r.lds_lane), "s"(r.src_delta), "s"(b_lo), "s"(b_hi)
                 : "vcc");
    return reinterpret_cast<const char *>(((size_t)hi << 32) | lo);
}

// r.lds_lane again, from nothing but the lane id.  It is the one per-lane value every unit of every layer needs (fragment reads, DMA
// source addresses), so it lives for the whole kernel -- and under the field kernel's register pressure hipcc parks it in scratch
// memory at three places; the reloads are vector-memory loads, and from then on its waitcnt pass puts `s_waitcnt vmcnt(0)` in front
// of the first use on every path a reload may have come from: right behind the hand-counted `vmcnt(8)` + barrier at the entry of
// EVERY layer (seen in the ISA of rounds 4-6: six full drains of the weight ring's DMAs per pass).  Redefining the value at each
// layer entry (three VALU instructions the compiler cannot hoist) ends the live range there: nothing to reload, nothing to wait for.
__device__ __forceinline__ void ring_refresh_lane(char *lds, Ring &r) {
    unsigned int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %0, -1, %0\n\t"
                 "v_lshl_add_u32 %0, %0, 4, %1"
                 : "=&v"(l)
                 : "s"((unsigned)(size_t)(const lds_char *)(lds + LDS_RING)));
    r.lds_lane = l;
}

template <int K>
__device__ __forceinline__ void ring_dma(const char *lane_src, char *dbase) {
    // address = this lane's source address of the slot (+ 4 KiB for the second group of four pieces) + immediate; the LDS
    // destination is wave-uniform; the instruction offset is added to the global AND to the LDS address
    // (LDS = M0 + offset + lane*16).  (The immediate is a 13-bit signed field: 4096 and up would silently wrap to
    // negative offsets.)
    __builtin_amdgcn_global_load_lds((glb_char *)(lane_src + (K / 4) * 4096), (lds_char *)(dbase + (K / 4) * 4096), 16, (K % 4) * 1024, 0);
}

__device__ __forceinline__ void ring_issue(char *lds, const Ring &r, int slot_global, int slot_in_pass) {
    const int pos = slot_global & (NSLOT - 1);
    const char *src = ring_lane_src(r.wbytes + (size_t)slot_in_pass * SLOT_BYTES, r);
    char *dbase = lds + LDS_RING + pos * SLOT_BYTES + r.wave * (PIECES * 1024);
    ring_dma<0>(src, dbase); ring_dma<1>(src, dbase); ring_dma<2>(src, dbase); ring_dma<3>(src, dbase);
    if constexpr (PIECES == 8) {
        ring_dma<4>(src, dbase); ring_dma<5>(src, dbase); ring_dma<6>(src, dbase); ring_dma<7>(src, dbase);
    }
}

// One of the 4 DMA pieces of the refill granted by the last ring_acquire.  They are issued one behind each of the next
// unit's first four MFMAs: a global_load_lds costs the issuing wave ~16 cycles of address processing, which fits in
// the shadow of a 32-cycle MFMA but was dead matrix time when all four followed the barrier back to back.
template <int K>
__device__ __forceinline__ void ring_issue_piece(char *lds, const Ring &r) {
    const int pos = r.pend_global & (NSLOT - 1);
    char *dbase = lds + LDS_RING + pos * SLOT_BYTES + r.wave * (PIECES * 1024);
    ring_dma<K>(r.pend_src, dbase);
}

// make slot r.g (and r.g+1) readable for everybody, free slot r.g-1 for its refill (ring_issue_piece<0..3>)
template <int DBG>
__device__ __forceinline__ int ring_acquire(char *lds, Ring &r) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DMA_AHEAD - 2) * PIECES) : "memory");
    if constexpr (!(DBG & 2)) __builtin_amdgcn_s_barrier();
    r.pend_global = r.g + DMA_AHEAD;
    r.pend_in_pass = r.next_in_pass;
    r.pend_src = ring_lane_src(r.wbytes + (size_t)r.next_in_pass * SLOT_BYTES, r);
    r.next_in_pass = r.next_in_pass + 1 == r.slots_per_pass ? 0 : r.next_in_pass + 1;
    const int pos = r.g & (NSLOT - 1);
    r.g++;
    return pos;
}

// The rest of this pass's weight stream is not needed (colour branch skipped): the DMA_AHEAD slots in flight hold its next
// layer.  Refill their ring positions with the first slots of the NEXT pass, exactly the state the kernel starts in.  Every
// wave owns its quarter of a slot for both the stale and the new pieces; the wait lets the stale ones land first (they were
// issued a whole layer ago: nothing is waited for in practice).  No barrier: nobody reads the positions being refilled, and the
// position of the last consumed slot (which slower waves may still be reading) is not touched.
__device__ __forceinline__ void ring_restart(char *lds, Ring &r) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int sl = 0; sl < DMA_AHEAD; sl++) ring_issue(lds, r, r.g + sl, sl);
    r.next_in_pass = DMA_AHEAD;
}

// Per-lane view of a 256-vector in the C/D register layout: element (IB, Q, e) is feature
// 32*IB + 16*Q + (e&3) + 8*(e>>2) + 4*h.
// LDS reads of the small constant tables are issued through inline asm: hipcc's waitcnt pass cannot tell them
// from reads of the DMA ring (same __shared__ array) and would otherwise put `s_waitcnt vmcnt(0)` in front of each
// one, draining the whole 7-slot DMA pipeline six times per layer (seen in the ISA; ~25 % of the kernel time).
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(const lds_char *)p;
}

// Weight-fragment reads (LDS ring -> registers) are inline asm with HAND-COUNTED waits.  Left to hipcc, every LDS
// wait behind a pending LDS-DMA becomes lgkmcnt(0) (219 of them per pass in the ISA of the previous version):
// each one also drains the prefetch issued a few instructions earlier and exposes a full LDS round trip.  The rule
// that makes the counts static: within a unit the 4 fragment reads (one behind each of the first 4 MFMAs) are the
// LAST LDS operations issued, so "lgkmcnt(4)" at the start of unit U means "everything issued before unit U-1's
// fragment reads has landed" = unit U's fragments (issued during unit U-2), its bias blocks and all older reads.
// tools/check_lds_hazards.py replays the compiled ISA and checks that no instruction reads a register whose
// ds_read has not been waited for.
template <int OFF>
__device__ __forceinline__ void ds_read16(half8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

template <int U_IN_SLOT>
__device__ __forceinline__ void lds_unit(const Ring &r, int pos, half8 (&a)[4]) {
    const unsigned q = r.lds_lane + pos * SLOT_BYTES;
    ds_read16<U_IN_SLOT * 4096>(a[0], q);          // (ib, hi)
    ds_read16<U_IN_SLOT * 4096 + 1024>(a[1], q);   // (ib, lo)
    ds_read16<U_IN_SLOT * 4096 + 2048>(a[2], q);   // (ib+1, hi)
    ds_read16<U_IN_SLOT * 4096 + 3072>(a[3], q);   // (ib+1, lo)
}

template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// 4 x 16 B at p, p+32, p+64, p+96 (bytes): the 16 accumulator-layout values of one 32-row block
__device__ __forceinline__ f32x16 lds_read_block(const float *p) {
    f32x4 v0, v1, v2, v3;
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:32\n\t"
        "ds_read_b128 %2, %4 offset:64\n\t"
        "ds_read_b128 %3, %4 offset:96\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(lds_addr(p))
        : "memory");
    f32x16 c;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        c[e] = v0[e]; c[4 + e] = v1[e]; c[8 + e] = v2[e]; c[12 + e] = v3[e];
    }
    return c;
}

// 2 x 16 B at p and p+32 (bytes)
__device__ __forceinline__ void lds_read_8(const float *p, float (&o)[8]) {
    f32x4 v0, v1;
    asm volatile(
        "ds_read_b128 %0, %2\n\t"
        "ds_read_b128 %1, %2 offset:32\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1)
        : "v"(lds_addr(p))
        : "memory");
#pragma unroll
    for (int e = 0; e < 4; e++) {
        o[e] = v0[e]; o[4 + e] = v1[e];
    }
}

__device__ __forceinline__ float vmax(float a, float b) {
    float r;   // plain v_max_f32: fmaxf() adds a canonicalising v_max in front of every operand
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// bias vector of row block IB in accumulator layout (it seeds the accumulator: first MFMA's C operand)
template <int IB>
__device__ __forceinline__ f32x16 bias_block(const float *bias, int h) {
    return lds_read_block(bias + 32 * IB + 4 * h);
}

// Activation of 4 accumulator values (half a B fragment), cut into 6 stages of <= 6 mutually INDEPENDENT VALU
// instructions.  With one wave per SIMD instructions issue in order: a VALU instruction behind an MFMA that waits
// for the matrix pipe waits too, and back-to-back dependent VALU instructions cost ~10 cycles each.  Putting one
// stage after each of a unit's 6 MFMAs (and fencing with sched_barrier) gives every MFMA gap ~5 independent
// instructions: the activation then costs no matrix time.  (Measured before this change: the same instructions,
// emitted as per-value dependent chains after the unit's last MFMA, took 9.2 ms of a 19.2 ms kernel.)
//   fragment T = 2*IB + Q of the next layer, HS = which 4 of its 8 elements:
//   value e is accumulator register 8*Q + 4*HS + e of row block IB = feature 32*IB + 16*Q + 8*HS + e + 4*h
// The layer bias is added HERE (one v_add per value) instead of seeding the accumulators: a seed costs 16
// v_accvgpr_write per row block plus an LDS read that has to be waited for right in front of the block's first MFMA.
// The 4 bias values (and, for fc_4, the 4 density-head weights) of a half fragment are fetched one unit ahead
// (ActIn), in front of that unit's fragment prefetches, so the unit-start wait covers them.
struct ActRegs {
    float x[4], y[4];
    fp16x2 hp[2], lp[2];
};

struct ActIn {
    f32x4 b;   // bias of the 4 features
    f32x4 w;   // density-head weights of the 4 features (SIG only)
};

// plain v_max_f32 (fmaxf / fmed3 get a canonicalising v_max in front of every operand)
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// write two packed f16 pairs into dwords 2*HS, 2*HS+1 of a fragment (whole-dword moves: 16-bit element inserts
// into a half8 are lowered through scratch memory by hipcc)
template <int HS>
__device__ __forceinline__ void put_pairs(half8 &frag, fp16x2 p0, fp16x2 p1) {
    u32x4v t = __builtin_bit_cast(u32x4v, frag);
    t[2 * HS] = __builtin_bit_cast(unsigned int, p0);
    t[2 * HS + 1] = __builtin_bit_cast(unsigned int, p1);
    frag = __builtin_bit_cast(half8, t);
}

// issue (no wait) the LDS reads of a half fragment's activation inputs
template <int T, int HS, bool SIG>
__device__ __forceinline__ void act_fetch(const float *bias, const float *wsig, int h, ActIn &in) {
    constexpr int F = 32 * (T / 2) + 16 * (T % 2) + 8 * HS;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(in.b) : "v"(lds_addr(bias + 4 * h)), "n"(F * 4));
    if constexpr (SIG) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(in.w) : "v"(lds_addr(wsig + 4 * h)), "n"(F * 4));
}

// LO = false: the consumer of this fragment is a 2-term layer (no Whi.Xlo product): only hi is produced
template <int T, int HS, bool SIG, int STAGE, bool LO = true>
__device__ __forceinline__ void act_stage(const f32x16 (&acc)[8], const ActIn &in, half8 (&bh)[16], half8 (&bl)[16],
                                          float &part, ActRegs &g, float k = 1.f) {
    constexpr int IB = T / 2, Q = T % 2;
    if constexpr (STAGE == 0) {
#pragma unroll
        for (int e = 0; e < 4; e++) g.y[e] = acc[IB][8 * Q + 4 * HS + e];                 // 4 x v_accvgpr_read
    } else if constexpr (STAGE == 1) {
#pragma unroll
        for (int e = 0; e < 4; e++) g.y[e] = __builtin_fmaf(g.y[e], k, in.b[e]);   // k == 1 (a literal): folds to v_add
    } else if constexpr (STAGE == 2) {
        // a' = 1.5 x + |x| = LeakyReLU_0.2(x) / 0.4 : ONE v_fma (|x| is a free source modifier); the 0.4 lives in
        // the next layer's packed weights and in the density-head weights
#pragma unroll
        for (int e = 0; e < 4; e++) g.x[e] = __builtin_fmaf(g.y[e], 1.5f, __builtin_fabsf(g.y[e]));
    } else if constexpr (STAGE == 3) {
        g.hp[0] = cvt_rtn(g.x[0], g.x[1]);
        g.hp[1] = cvt_rtn(g.x[2], g.x[3]);
        if constexpr (SIG) part += in.w[0] * g.x[0] + in.w[1] * g.x[1] + in.w[2] * g.x[2] + in.w[3] * g.x[3];
    } else if constexpr (STAGE == 4) {
        if constexpr (!LO) return;
        // remainder x - float(hi) in one v_fma_mix_f32 per value (reads the f16 half directly)
        const unsigned int p0 = __builtin_bit_cast(unsigned int, g.hp[0]), p1 = __builtin_bit_cast(unsigned int, g.hp[1]);
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(g.y[0]) : "v"(p0), "v"(g.x[0]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(g.y[1]) : "v"(p0), "v"(g.x[1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(g.y[2]) : "v"(p1), "v"(g.x[2]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(g.y[3]) : "v"(p1), "v"(g.x[3]));
    } else {
        put_pairs<HS>(bh[T], g.hp[0], g.hp[1]);
        if constexpr (LO) {
            g.lp[0] = cvt_rtn(g.y[0], g.y[1]);
            g.lp[1] = cvt_rtn(g.y[2], g.y[3]);
            put_pairs<HS>(bl[T], g.lp[0], g.lp[1]);
        }
    }
}

// whole half-fragment at once (used where nothing can hide it: the tail of the first layer)
template <int T, int HS, bool SIG>
__device__ __forceinline__ void act_half(const f32x16 (&acc)[8], const float *bias, const float *wsig, int h, half8 (&bh)[16],
                                         half8 (&bl)[16], float &part, float k = 1.f) {
    ActRegs g;
    ActIn in;
    act_fetch<T, HS, SIG>(bias, wsig, h, in);
    if constexpr (SIG) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(in.b), "+v"(in.w)::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(in.b)::"memory");
    act_stage<T, HS, SIG, 0>(acc, in, bh, bl, part, g);
    act_stage<T, HS, SIG, 1>(acc, in, bh, bl, part, g, k);
    act_stage<T, HS, SIG, 2>(acc, in, bh, bl, part, g);
    act_stage<T, HS, SIG, 3>(acc, in, bh, bl, part, g);
    act_stage<T, HS, SIG, 4>(acc, in, bh, bl, part, g);
    act_stage<T, HS, SIG, 5>(acc, in, bh, bl, part, g);
}

template <int T, bool SIG>
__device__ __forceinline__ void act_step(const f32x16 (&acc)[8], const float *bias, const float *wsig, int h, half8 (&bh)[16],
                                         half8 (&bl)[16], float &part, float k = 1.f) {
    act_half<T, 0, SIG>(acc, bias, wsig, h, bh, bl, part, k);
    act_half<T, 1, SIG>(acc, bias, wsig, h, bh, bl, part, k);
}

// The density head's contribution of a finished layer's LOWER half (fragments 8..15 = accumulators acc[4..7]) without producing
// the fragments: stages 0..3 of act_stage for half fragment J -- the very functions, in the very order, the next layer's pending
// work runs later, so the sum is bit-identical to the one that layer accumulates.  mlp_kernel uses it to know sigma of a pass
// BEFORE the colour layers start (colour-branch skipping).  The LDS reads of the bias / weight rows run one half fragment ahead.
template <int J>
__device__ __forceinline__ void sigma_half(const f32x16 (&acc)[8], const float *bias, const float *wsig, int h, half8 (&bh)[16],
                                           half8 (&bl)[16], ActIn (&in)[2], float &part, float k) {
    constexpr int T = 8 + J / 2, HS = J % 2;
    if constexpr (J + 1 < 16) {
        act_fetch<8 + (J + 1) / 2, (J + 1) % 2, true>(bias, wsig, h, in[(J + 1) & 1]);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(in[J & 1].b), "+v"(in[J & 1].w)::"memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(in[J & 1].b), "+v"(in[J & 1].w)::"memory");
    }
    ActRegs g;
    act_stage<T, HS, true, 0>(acc, in[J & 1], bh, bl, part, g);
    act_stage<T, HS, true, 1>(acc, in[J & 1], bh, bl, part, g, k);
    act_stage<T, HS, true, 2>(acc, in[J & 1], bh, bl, part, g);
    act_stage<T, HS, true, 3>(acc, in[J & 1], bh, bl, part, g);
}

template <int... Js>
__device__ __forceinline__ void sigma_lower_half(std::integer_sequence<int, Js...>, const f32x16 (&acc)[8], const float *bias,
                                                 const float *wsig, int h, half8 (&bh)[16], half8 (&bl)[16], float &part, float k) {
    ActIn in[2];
    act_fetch<8, 0, true>(bias, wsig, h, in[0]);
    (sigma_half<Js>(acc, bias, wsig, h, bh, bl, in, part, k), ...);
}

// One 8-row-block layer (NS k-steps) from the LDS ring.
//   pend:  activation of the PREVIOUS layer's lower half (row blocks 4-7 -> B fragments 8..15, bias_pend), one
//          half fragment per unit from unit 0 (HAS_PEND), hidden behind this layer's first MFMAs;
//   own:   activation of this layer's upper half into B fragments 0 .. NS/2-1 during the last NS/2 k-steps of
//          the lower half.
// On return acc[0..3] are consumed (except fragments t >= NS/2 when NS < 16), acc[4..7] hold the lower half.
// Every index below is a compile-time constant (template recursion over the unit number): register arrays must
// never be indexed dynamically or they end up in scratch memory.
constexpr int RING_DEPTH = 3;   // register ring of fragment units: 2 units (384 matrix cycles) ahead of the MFMAs

struct LayerState {
    half8 ring[RING_DEPTH][4];
    ActIn in[2];                // activation inputs of the half fragment that STARTS in unit U, fetched during unit U-1 (ActPlan::SLOT)
    ActRegs g;                  // a half fragment in flight across two units (ActPlan::PHASE 1 -> 2)
    int pos_cur, pos_nxt;
};

// one LDS read of the register ring's next unit (fragment F of unit U_IN_SLOT of ring position pos)
template <int U_IN_SLOT, int F>
__device__ __forceinline__ void lds_frag(const Ring &r, int pos, half8 &dst) {
    ds_read16<U_IN_SLOT * 4096 + F * 1024>(dst, r.lds_lane + pos * SLOT_BYTES);
}

// which activation work is hidden in unit U of a layer8
//
// What one in-order wave per SIMD can hide behind a unit's MFMAs is bounded by ISSUE time, not by the matrix pipe
// (tools/unit_cost_ubench.hip): a unit of 6 MFMAs with its 5 LDS reads hides ~18 VALU instructions; the ~36 of a whole half
// fragment make it 267 cycles instead of 201.  So a 16-k-step layer whose activation work has 60 units to go to spreads it
// (SPREAD): a half fragment takes TWO units -- stages 0..2 in one (PHASE 1), 3..5 in the next (PHASE 2), its registers kept
// in LayerState::g -- except the two that the deadlines leave one unit for (PHASE 0, all six stages as before):
//   pending (previous layer's lower half -> fragments 8..15; fragment 8+k is first needed at unit 16+2k):
//       half fragment j = 0, 1 in units 0, 1;  j >= 2 in units 2j-2, 2j-1  (j = 15: units 28, 29 < 30)
//   own (this layer's upper half -> fragments 0..7; fragment T is free from unit 34+2T on):
//       half fragment j <= 13 in units 34+2j, 35+2j (fragment T from unit 34+4T);  j = 14, 15 in units 62, 63
template <int DBG, int NS, bool HAS_PEND, bool SIG_PEND, bool SIG_OWN, int U, bool SPREAD = false>
struct ActPlan {
    static constexpr bool IN_RANGE = U >= 0 && U < NS * 4;
    static constexpr int HALF = U / (2 * NS), REM = U % (2 * NS), S = REM >> 1;
    // the previous layer's lower half (fragments 8..15, half a fragment per unit over units 0..15: fragment 8+k
    // is first needed at unit 16+2k), or this layer's upper half (fragments 0..NS/2-1) during the last NS/2 k-steps
    // of the lower half
    static constexpr bool PEND = IN_RANGE && HAS_PEND && U < (SPREAD ? 30 : 16) && !(DBG & 4);
    // own: fragment T may be overwritten once k-step T has been consumed by both row blocks of this half, i.e. from
    // k-step T+1 on.  NS == 16: fragments 0..7 during k-steps 8..15; NS < 16 (first layers): fragments 0..NS-2 during
    // k-steps 1..NS-1 (the last upper-half fragments are activated after the layer, un-hidden)
    static constexpr int S0 = NS == 16 ? 8 : 1;
    static constexpr bool OWN = IN_RANGE && (SPREAD ? U >= 34 : (HALF == 1 && S >= S0)) && !(DBG & 4);
    static constexpr int M = U - 2 * NS - 2 * S0;
    // half fragment index within its group and the part of it done in this unit
    static constexpr int J = !SPREAD ? (PEND ? U : (OWN ? M : 0))
                             : PEND ? (U < 2 ? U : 2 + (U - 2) / 2)
                             : OWN  ? (U >= 62 ? 14 + (U - 62) : (U - 34) / 2) : 0;
    static constexpr int PHASE = !SPREAD ? 0 : PEND ? (U < 2 ? 0 : 1 + (U - 2) % 2) : OWN ? (U >= 62 ? 0 : 1 + (U - 34) % 2) : 0;
    static constexpr int T = PEND ? 8 + J / 2 : (OWN ? J / 2 : 0);
    static constexpr int HS = J % 2;
    static constexpr bool SIG = PEND ? SIG_PEND : SIG_OWN;
    static constexpr bool ACT = PEND || OWN;
    static constexpr bool STARTS = ACT && PHASE != 2;       // its inputs are fetched one unit earlier ...
    static constexpr int SLOT = SPREAD ? J % 2 : U % 2;     // ... into this ActIn (consecutive half fragments alternate)
    // activation stage (0..5, or -1) behind MFMA K (0..5) of this unit; a half-rate unit uses the two gaps without
    // fragment reads and one of the others
    static constexpr int stage(int K) {
        return !ACT ? -1 : PHASE == 0 ? K : (K == 2 ? 0 : K == 4 ? 1 : K == 5 ? 2 : -4) + (PHASE == 2 ? 3 : 0);
    }
};

// The deadlines the schedules above rest on, checked at compile time for both plans of a 16-k-step layer: every half fragment
// of the pending group is finished before the unit that first multiplies with its fragment (16 + 2k for fragment 8 + k), no half
// fragment of the own group starts before its fragment's last use (k-step T of the lower half: units 32 + 2T, 33 + 2T), each is
// visited once per phase in consecutive units, fragments 0..3 are complete before unit 56 (where a layer feeding an MX layer
// converts K block 0), and consecutive half fragments alternate between the two ActIn slots.
template <bool SPREAD, int... Us>
constexpr bool act_plan_ok(std::integer_sequence<int, Us...>) {
    int first[2][16] = {}, last[2][16] = {}, visits[2][16] = {}, slot[2][16] = {};
    for (int g = 0; g < 2; g++)
        for (int k = 0; k < 16; k++) first[g][k] = last[g][k] = -1;
    bool ok = true;
    auto visit = [&](int U, bool pend, bool own, int j, int phase, int sl, bool starts) {
        if (!pend && !own) return;
        if (pend && own) ok = false;
        const int g = own ? 1 : 0;
        if (first[g][j] < 0) { first[g][j] = U; slot[g][j] = sl; if (!starts || phase == 2) ok = false; }
        else if (U != last[g][j] + 1 || phase != 2 || starts || sl != slot[g][j]) ok = false;
        last[g][j] = U;
        visits[g][j]++;
    };
    (visit(Us, ActPlan<0, 16, true, false, false, Us, SPREAD>::PEND, ActPlan<0, 16, true, false, false, Us, SPREAD>::OWN,
           ActPlan<0, 16, true, false, false, Us, SPREAD>::J, ActPlan<0, 16, true, false, false, Us, SPREAD>::PHASE,
           ActPlan<0, 16, true, false, false, Us, SPREAD>::SLOT, ActPlan<0, 16, true, false, false, Us, SPREAD>::STARTS), ...);
    for (int j = 0; j < 16; j++) {
        const int k = j / 2;
        if (visits[0][j] != (last[0][j] - first[0][j] + 1) || visits[1][j] != (last[1][j] - first[1][j] + 1)) ok = false;
        if (first[0][j] < 0 || last[0][j] >= 16 + 2 * k) ok = false;                        // pending: fragment 8 + k ready in time
        if (first[1][j] < 34 + 2 * k || last[1][j] > 63) ok = false;                        // own: fragment k free, done inside the layer
        if (k < 4 && last[1][j] >= 56) ok = false;                                          // K block 0 complete before its conversion
        if (j > 0 && (slot[0][j] == slot[0][j - 1] || slot[1][j] == slot[1][j - 1])) ok = false;
    }
    return ok && slot[1][0] != slot[0][15];
}
static_assert(act_plan_ok<false>(std::make_integer_sequence<int, 64>{}), "ActPlan: one half fragment per unit");
static_assert(act_plan_ok<true>(std::make_integer_sequence<int, 64>{}), "ActPlan<SPREAD>: half-rate schedule");

template <int DBG, int NS, bool HAS_PEND, bool SIG_PEND, bool SIG_OWN, int U, bool SPREAD = false>
__device__ __forceinline__ void layer8_fetch(const float *bias, const float *bias_pend, const float *wsig, int h, LayerState &st) {
    using P = ActPlan<DBG, NS, HAS_PEND, SIG_PEND, SIG_OWN, U, SPREAD>;
    if constexpr (P::STARTS) act_fetch<P::T, P::HS, P::SIG>(P::PEND ? bias_pend : bias, wsig, h, st.in[P::SLOT]);
}

// TERMS = 3: Whi.Xhi + Wlo.Xhi + Whi.Xlo (6 MFMAs per unit);  TERMS = 2: the Whi.Xlo products are dropped (4 MFMAs per
// unit; the colour layers fc_5 / fc_6, whose error is not amplified by the density head -- DESIGN.md).
// LO_PEND / LO_OWN: whether the fragments activated in this layer (previous layer's lower half / this layer's upper
// half) need their lo part, i.e. whether their CONSUMER is a 3-term layer.
// the layers whose activation work is spread at half rate (ActPlan): 16 k-steps, 6 MFMAs per unit, pending work
constexpr bool layer8_spread(int NS, bool HAS_PEND, int TERMS) { return NS == 16 && HAS_PEND && TERMS == 3; }

template <int DBG, int NS, bool HAS_PEND, bool SIG_PEND, bool SIG_OWN, int TERMS, bool LO_PEND, bool LO_OWN, int U>
__device__ __forceinline__ void layer8_unit(char *lds, Ring &r, LayerState &st, half8 (&bh)[16], half8 (&bl)[16],
                                            f32x16 (&acc)[8], const float *bias, const float *bias_pend, const float *wsig,
                                            int h, float &part, float k_own, float k_pend) {
    constexpr int UNITS = NS * 4, RD = RING_DEPTH, UPS = UNITS_PER_SLOT;
    constexpr bool SPREAD = layer8_spread(NS, HAS_PEND, TERMS) && !(DBG & 16);
    using P = ActPlan<DBG, NS, HAS_PEND, SIG_PEND, SIG_OWN, U, SPREAD>;
    if constexpr (U % UPS == 0 && U != 0) {
        st.pos_cur = ring_acquire<DBG>(lds, r);
        st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    }
    constexpr int UN = U + RD - 1;
    constexpr bool PF = UN < UNITS && !(DBG & 8);
    const int pf_pos = (UN / UPS) == (U / UPS) ? st.pos_cur : st.pos_nxt;
    constexpr int S = P::S, IB = 4 * P::HALF + 2 * (P::REM & 1);
    constexpr int T = P::T, HS = P::HS;
    constexpr bool SIG = P::SIG, ACT = P::ACT;
    constexpr bool LO = P::PEND ? LO_PEND : LO_OWN;
    ActRegs g_unit;
    ActRegs &g = SPREAD ? st.g : g_unit;
    half8(&a)[4] = st.ring[U % RD];
    half8(&nx)[4] = st.ring[UN % RD];
    const ActIn &in = st.in[P::SLOT];
    // this unit's fragments (issued during unit U-2) and activation inputs (issued at the start of unit U-1) have
    // landed once only unit U-1's 4 fragment reads are outstanding
    constexpr bool PF_PREV = U == 0 || ((U - 1 + RD - 1) < UNITS && !(DBG & 8));
    lds_wait<PF_PREV ? 4 : 0>();
    layer8_fetch<DBG, NS, HAS_PEND, SIG_PEND, SIG_OWN, U + 1, SPREAD>(bias, bias_pend, wsig, h, st);
#define SDN_STAGE(K) \
    if constexpr (U % UPS < PIECES / 4 && K < 4 && !(DBG & 1)) ring_issue_piece<4 * (U % UPS) + ((K) & 3)>(lds, r); \
    if constexpr (P::stage(K) >= 0) act_stage<T, HS, SIG, P::stage(K) < 0 ? 0 : P::stage(K), LO>(acc, in, bh, bl, part, g, P::PEND ? k_pend : k_own); \
    if constexpr (PF && K < 4) lds_frag<UN % UPS, (K) & 3>(r, pf_pos, nx[(K) & 3]); \
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG & 16) {
        asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(bh[S]), "v"(bl[S]));
        if constexpr (S == 0) { acc[IB] = zero16(); acc[IB + 1] = zero16(); }
        SDN_STAGE(0) SDN_STAGE(1) SDN_STAGE(2) SDN_STAGE(3) SDN_STAGE(4) SDN_STAGE(5)
    } else if constexpr (TERMS == 2) {
        // 4 MFMAs: the six activation stages share four gaps (stage 4 is empty and stage 5 a single move when !LO)
        if constexpr (S == 0) acc[IB] = mfma16(a[0], bh[S], zero16());
        else acc[IB] = mfma16(a[0], bh[S], acc[IB]);
        SDN_STAGE(0) SDN_STAGE(1)
        if constexpr (S == 0) acc[IB + 1] = mfma16(a[2], bh[S], zero16());
        else acc[IB + 1] = mfma16(a[2], bh[S], acc[IB + 1]);
        SDN_STAGE(2)
        acc[IB] = mfma16(a[1], bh[S], acc[IB]);
        SDN_STAGE(3)
        acc[IB + 1] = mfma16(a[3], bh[S], acc[IB + 1]);
        SDN_STAGE(4) SDN_STAGE(5)
    } else {
        if constexpr (S == 0) acc[IB] = mfma16(a[0], bh[S], zero16());
        else acc[IB] = mfma16(a[0], bh[S], acc[IB]);
        SDN_STAGE(0)
        if constexpr (S == 0) acc[IB + 1] = mfma16(a[2], bh[S], zero16());
        else acc[IB + 1] = mfma16(a[2], bh[S], acc[IB + 1]);
        SDN_STAGE(1)
        acc[IB] = mfma16(a[1], bh[S], acc[IB]);
        SDN_STAGE(2)
        acc[IB + 1] = mfma16(a[3], bh[S], acc[IB + 1]);
        SDN_STAGE(3)
        acc[IB] = mfma16(a[0], bl[S], acc[IB]);
        SDN_STAGE(4)
        acc[IB + 1] = mfma16(a[2], bl[S], acc[IB + 1]);
        SDN_STAGE(5)
    }
#undef SDN_STAGE
}

template <int DBG, int NS, bool HAS_PEND, bool SIG_PEND, bool SIG_OWN, int TERMS, bool LO_PEND, bool LO_OWN, int... Us>
__device__ __forceinline__ void layer8_units(std::integer_sequence<int, Us...>, char *lds, Ring &r, LayerState &st,
                                             half8 (&bh)[16], half8 (&bl)[16], f32x16 (&acc)[8], const float *bias,
                                             const float *bias_pend, const float *wsig, int h, float &part, float k_own,
                                             float k_pend) {
    (layer8_unit<DBG, NS, HAS_PEND, SIG_PEND, SIG_OWN, TERMS, LO_PEND, LO_OWN, Us>(lds, r, st, bh, bl, acc, bias, bias_pend, wsig, h, part,
                                                                                   k_own, k_pend), ...);
}

template <int DBG, int NS, bool HAS_PEND, bool SIG_PEND, bool SIG_OWN, int TERMS = 3, bool LO_PEND = true, bool LO_OWN = true>
__device__ __forceinline__ void layer8(char *lds, Ring &r, half8 (&bh)[16], half8 (&bl)[16], f32x16 (&acc)[8],
                                       const float *bias, const float *bias_pend, const float *wsig, int h, float &part,
                                       float k_own = 1.f, float k_pend = 1.f) {
    LayerState st;
    ring_refresh_lane(lds, r);
    st.pos_cur = ring_acquire<DBG>(lds, r);
    st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    layer8_fetch<DBG, NS, HAS_PEND, SIG_PEND, SIG_OWN, 0, layer8_spread(NS, HAS_PEND, TERMS) && !(DBG & 16)>(bias, bias_pend, wsig, h, st);
    lds_unit<0>(r, st.pos_cur, st.ring[0]);
    lds_unit<1>(r, st.pos_cur, st.ring[1]);
    layer8_units<DBG, NS, HAS_PEND, SIG_PEND, SIG_OWN, TERMS, LO_PEND, LO_OWN>(std::make_integer_sequence<int, NS * 4>{}, lds, r, st, bh,
                                                                             bl, acc, bias, bias_pend, wsig, h, part, k_own, k_pend);
}

// =====================================================================================================
// Colour layers as  Whi.Xhi (f16)  +  block-scaled fp6 corrections  [Wlo | Whi] . [X ; Xlo]
// =====================================================================================================
// The two correction terms of the 3-term split only need ~5 significant bits (their sum is 2^-11 of the product), so in
// the layers whose error nothing amplifies (fc_5, fc_6: the colour branch) they are evaluated with
// v_mfma_scale_f32_32x32x64_f8f6f4 on fp6 (e2m3) operands: K = 64 per instruction at the issue cost of one K = 16 f16
// MFMA.  192 MFMAs per layer instead of 384; measured error of the emulation (tools/precision_study.py) 4e-5 on net_out
// against 5-7e-4 for simply dropping a term.  Operand facts (pinned on the hardware by tools/mx_probe.hip): lane l holds
// row / column l & 31 and the 32 k values 32 * (l >> 5) + i as 6-bit fields, little endian, in 6 dwords; the E8M0 scale
// byte (2^(b - 127)) of the lane's 32-value block comes from byte OPSEL of a per-lane VGPR;
// v_cvt_scalef32_pk32_fp6_f16 converts 32 f16 (16 VGPRs, element p -> field p) dividing by a power-of-two scale, round to
// nearest even, saturating at 7.5.
//   B operands: K block kb = features 64 kb .. 64 kb + 63 = B fragments 4 kb .. 4 kb + 3; a lane's 32 values are its 8
//   elements of each of the 4 fragments (the cvt instruction reads the 16 VGPRs of bh[4kb .. 4kb+3] / bl[...] as they are).
//   Per-lane scale: biased exponent of the block's max |x| minus 2 (max lands in [4, 8): at most the top value saturates),
//   the lo block uses that exponent minus 11 (|x - f16(x)| <= 2^-11 of x's binade).
//   A operands: packed by pack_mx_kernel with one scale per row and 32-k block.
typedef unsigned int u32x6v __attribute__((ext_vector_type(6)));
typedef int i32x8v __attribute__((ext_vector_type(8)));
typedef _Float16 half32 __attribute__((ext_vector_type(32)));

struct MxState {
    u32x6v x6[4], xl6[4];   // fp6 images of the hi / lo f16 fragments of K block kb
    int sx[4];              // byte 0: scale of x6[kb], byte 1: scale of xl6[kb]
    float bm[4];            // running max |x| of K block kb (reset by mx_convert)
};
// (Keeping the fp6 images in the registers of the lo fragments they replace -- bl[4kb .. 4kb+3] are dead once converted --
// was tried: hipcc then fuses the reads of neighbouring fragments into 32-byte loads of the fragment array, which sends
// the array to scratch memory; with that blocked, the contiguity constraints cost more moves and spills than the 52
// extra registers of this struct.)

__device__ __forceinline__ half32 cat4(const half8 &a, const half8 &b, const half8 &c, const half8 &d) {
    const auto ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const auto cd = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    return __builtin_shufflevector(ab, cd, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25,
                                   26, 27, 28, 29, 30, 31);
}

template <int KB>
__device__ __forceinline__ void mx_convert(const half8 (&bh)[16], const half8 (&bl)[16], MxState &mx) {
    int e = (int)((__builtin_bit_cast(unsigned int, mx.bm[KB]) >> 23) & 255u) - 2;
    e = e < 12 ? 12 : e;                                   // e - 11 stays a normal scale; such blocks are ~0 anyway
    const float s_hi = __builtin_bit_cast(float, (unsigned int)e << 23);
    const float s_lo = __builtin_bit_cast(float, (unsigned int)(e - 11) << 23);
    mx.x6[KB] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(cat4(bh[4 * KB], bh[4 * KB + 1], bh[4 * KB + 2], bh[4 * KB + 3]), s_hi);
    mx.xl6[KB] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(cat4(bl[4 * KB], bl[4 * KB + 1], bl[4 * KB + 2], bl[4 * KB + 3]), s_lo);
    mx.sx[KB] = e | ((e - 11) << 8);
    mx.bm[KB] = 0.f;
}

// one fp6 MFMA: A = the two 16-byte ring fragments of an fp6 weight fragment (6 dwords of fields, scale word, pad),
// B = a fp6 activation block, OPB = which byte of sb is its scale
template <int OPB>
__device__ __forceinline__ f32x16 mfma_mx(const half8 &a_lo, const half8 &a_hi, const u32x6v &b, f32x16 c, int sb) {
    // every dword of a ring fragment must stay allocated until its (asynchronous, hand-waited) ds_read has landed: the
    // pad dword of a_hi is not an MFMA operand, so it is named here -- otherwise hipcc reuses that register as a
    // temporary right behind the read's issue and the data landing later overwrites it (found the hard way:
    // tools/check_lds_hazards.py reports exactly this)
    asm volatile("" ::"v"(a_hi));
    const u32x4v w0 = __builtin_bit_cast(u32x4v, a_lo), w1 = __builtin_bit_cast(u32x4v, a_hi);
    // (a_hi's two code dwords are copied behind a_lo by two v_mov in front of every fp6 MFMA: the 6-register operand cannot
    // overlap a 4-register fragment partially, whatever the vector is built from -- tried)
    const i32x8v A = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], 0, 0};
    const i32x8v B = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], (int)b[4], (int)b[5], 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 2, 2, 0, (int)w1[2], OPB, sb);
}

// running block max of K block KB taken from the finished f16 hi fragments (used where the activations were not produced by
// act_stage_x: the sky MLP's first layer).  |x| as 15-bit patterns order like the values; only the exponent is used.
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
template <int KB>
__device__ __forceinline__ void mx_block_max_f16(const half8 (&bh)[16], MxState &mx) {
    u16x2v m = {0, 0};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const u32x4v w = __builtin_bit_cast(u32x4v, bh[4 * KB + t]);
#pragma unroll
        for (int k = 0; k < 4; k++) m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2v, w[k] & 0x7fff7fffu));
    }
    const unsigned int top = m[0] > m[1] ? m[0] : m[1];
    mx.bm[KB] = __builtin_bit_cast(float, ((top >> 10) + 112u) << 23);   // 2^(exponent of the largest |x|)
}

// act_stage + running block max (the f32 activations of stage 2 are at hand in stage 3)
template <int T, int HS, bool SIG, int STAGE, bool MXT>
__device__ __forceinline__ void act_stage_x(const f32x16 (&acc)[8], const ActIn &in, half8 (&bh)[16], half8 (&bl)[16], MxState &mx,
                                            float &part, ActRegs &g, float k = 1.f) {
    act_stage<T, HS, SIG, STAGE, true>(acc, in, bh, bl, part, g, k);
    if constexpr (MXT && STAGE == 3) {
        float m = mx.bm[T / 4];
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(g.x[0]), "v"(g.x[1]));
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(g.x[2]), "v"(g.x[3]));
        mx.bm[T / 4] = m;
    }
}

// KIND 0: an ordinary 3-term layer whose OUTPUT feeds an MX layer (fc_4): its upper half is activated into f16 hi/lo
//         fragments 0..7 as always, K block 0 is converted at unit 56, K block 1 by the consumer's unit 0.
// KIND 1: MX layer fed by and feeding MX (fc_5).   KIND 2: MX layer whose output feeds a 3-term layer (fc_6 -> fc_out_c).
// MX unit order (pack_mx_kernel): per output half (row blocks 4*half .. +3) and K block kb, 8 units of 4 KiB:
//   0..3  f16 fragments of k-step 4 kb + t for the half's 4 row blocks                         -> 4 MFMAs
//   4, 5  fp6 Wlo fragments of row blocks (0,1) / (2,3) of the half  x  x6[kb]                  -> 2 MFMAs each
//   6, 7  fp6 Whi fragments of row blocks (0,1) / (2,3)              x  xl6[kb]                 -> 2 MFMAs each
// The activation schedule (which half fragment is activated behind which unit) is layer8's.
template <int DBG, int KIND, bool SIG_PEND, bool SIG_OWN, int U>
__device__ __forceinline__ void layer8x_unit(char *lds, Ring &r, LayerState &st, half8 (&bh)[16], half8 (&bl)[16], MxState &mx,
                                             f32x16 (&acc)[8], const float *bias, const float *bias_pend, const float *wsig,
                                             int h, float &part, float k_own, float k_pend) {
    constexpr int NS = 16, UNITS = 64, RD = RING_DEPTH, UPS = UNITS_PER_SLOT;
    constexpr bool MXL = KIND != 0;
    constexpr bool SPREAD = !MXL;   // MX units (4 or 2 MFMAs) are issue-bound wherever the activation work goes
    using P = ActPlan<DBG, NS, true, SIG_PEND, SIG_OWN, U, SPREAD>;
    if constexpr (U % UPS == 0 && U != 0) {
        st.pos_cur = ring_acquire<DBG>(lds, r);
        st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    }
    constexpr int UN = U + RD - 1;
    constexpr bool PF = UN < UNITS && !(DBG & 8);
    const int pf_pos = (UN / UPS) == (U / UPS) ? st.pos_cur : st.pos_nxt;
    constexpr int T = P::T, HS = P::HS;
    constexpr bool SIG = P::SIG, ACT = P::ACT;
    // does the fragment activated here feed an MX layer?  PEND fragments (8..15) feed THIS layer, OWN fragments the next
    constexpr bool MXT = P::PEND ? MXL : KIND != 2;
    // K blocks completed by the previous unit: converted in this unit's first gap
    constexpr int CONV = (MXL && U == 0) ? 1 : (MXL && U == 8) ? 2 : (MXL && U == 16) ? 3 : (KIND != 2 && U == 56) ? 0 : -1;
    ActRegs g_unit;
    ActRegs &g = SPREAD ? st.g : g_unit;
    half8(&a)[4] = st.ring[U % RD];
    half8(&nx)[4] = st.ring[UN % RD];
    const ActIn &in = st.in[P::SLOT];
    constexpr bool PF_PREV = U == 0 || ((U - 1 + RD - 1) < UNITS && !(DBG & 8));
    lds_wait<PF_PREV ? 4 : 0>();
    layer8_fetch<DBG, NS, true, SIG_PEND, SIG_OWN, U + 1, SPREAD>(bias, bias_pend, wsig, h, st);
#define SDN_STAGE(K) \
    if constexpr (U % UPS < PIECES / 4 && K < 4 && !(DBG & 1)) ring_issue_piece<4 * (U % UPS) + ((K) & 3)>(lds, r); \
    if constexpr (CONV >= 0 && K == 0) mx_convert<CONV < 0 ? 0 : CONV>(bh, bl, mx); \
    if constexpr (P::stage(K) >= 0) act_stage_x<T, HS, SIG, P::stage(K) < 0 ? 0 : P::stage(K), MXT>(acc, in, bh, bl, mx, part, g, P::PEND ? k_pend : k_own); \
    if constexpr (PF && K < 4) lds_frag<UN % UPS, (K) & 3>(r, pf_pos, nx[(K) & 3]); \
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!MXL) {
        constexpr int S = P::S, IB = 4 * P::HALF + 2 * (P::REM & 1);
        if constexpr (S == 0) acc[IB] = mfma16(a[0], bh[S], zero16());
        else acc[IB] = mfma16(a[0], bh[S], acc[IB]);
        SDN_STAGE(0)
        if constexpr (S == 0) acc[IB + 1] = mfma16(a[2], bh[S], zero16());
        else acc[IB + 1] = mfma16(a[2], bh[S], acc[IB + 1]);
        SDN_STAGE(1)
        acc[IB] = mfma16(a[1], bh[S], acc[IB]);
        SDN_STAGE(2)
        acc[IB + 1] = mfma16(a[3], bh[S], acc[IB + 1]);
        SDN_STAGE(3)
        acc[IB] = mfma16(a[0], bl[S], acc[IB]);
        SDN_STAGE(4)
        acc[IB + 1] = mfma16(a[2], bl[S], acc[IB + 1]);
        SDN_STAGE(5)
    } else {
        constexpr int HALF = U / 32, KB = (U % 32) / 8, SUB = U % 8, IB0 = 4 * HALF;
        if constexpr (SUB < 4) {
            constexpr int S = 4 * KB + SUB;
            if constexpr (S == 0) acc[IB0] = mfma16(a[0], bh[S], zero16());
            else acc[IB0] = mfma16(a[0], bh[S], acc[IB0]);
            SDN_STAGE(0) SDN_STAGE(1)
            if constexpr (S == 0) acc[IB0 + 1] = mfma16(a[1], bh[S], zero16());
            else acc[IB0 + 1] = mfma16(a[1], bh[S], acc[IB0 + 1]);
            SDN_STAGE(2)
            if constexpr (S == 0) acc[IB0 + 2] = mfma16(a[2], bh[S], zero16());
            else acc[IB0 + 2] = mfma16(a[2], bh[S], acc[IB0 + 2]);
            SDN_STAGE(3)
            if constexpr (S == 0) acc[IB0 + 3] = mfma16(a[3], bh[S], zero16());
            else acc[IB0 + 3] = mfma16(a[3], bh[S], acc[IB0 + 3]);
            SDN_STAGE(4) SDN_STAGE(5)
        } else {
            constexpr int TERM = (SUB - 4) / 2, IBA = IB0 + 2 * ((SUB - 4) % 2);
            constexpr bool ON = !(DBG & (TERM == 0 ? 32 : 64));   // ablation: DBG & 32 drops Wlo.X, DBG & 64 drops Whi.Xlo
            if constexpr (!ON) asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
            if constexpr (ON && TERM == 0) acc[IBA] = mfma_mx<0>(a[0], a[1], mx.x6[KB], acc[IBA], mx.sx[KB]);
            if constexpr (ON && TERM == 1) acc[IBA] = mfma_mx<1>(a[0], a[1], mx.xl6[KB], acc[IBA], mx.sx[KB]);
            SDN_STAGE(0) SDN_STAGE(1) SDN_STAGE(2)
            if constexpr (ON && TERM == 0) acc[IBA + 1] = mfma_mx<0>(a[2], a[3], mx.x6[KB], acc[IBA + 1], mx.sx[KB]);
            if constexpr (ON && TERM == 1) acc[IBA + 1] = mfma_mx<1>(a[2], a[3], mx.xl6[KB], acc[IBA + 1], mx.sx[KB]);
            SDN_STAGE(3) SDN_STAGE(4) SDN_STAGE(5)
        }
    }
#undef SDN_STAGE
}

template <int DBG, int KIND, bool SIG_PEND, bool SIG_OWN, int... Us>
__device__ __forceinline__ void layer8x_units(std::integer_sequence<int, Us...>, char *lds, Ring &r, LayerState &st, half8 (&bh)[16],
                                              half8 (&bl)[16], MxState &mx, f32x16 (&acc)[8], const float *bias,
                                              const float *bias_pend, const float *wsig, int h, float &part, float k_own,
                                              float k_pend) {
    (layer8x_unit<DBG, KIND, SIG_PEND, SIG_OWN, Us>(lds, r, st, bh, bl, mx, acc, bias, bias_pend, wsig, h, part, k_own, k_pend), ...);
}

template <int DBG, int KIND, bool SIG_PEND, bool SIG_OWN>
__device__ __forceinline__ void layer8x(char *lds, Ring &r, half8 (&bh)[16], half8 (&bl)[16], MxState &mx, f32x16 (&acc)[8],
                                        const float *bias, const float *bias_pend, const float *wsig, int h, float &part,
                                        float k_own = 1.f, float k_pend = 1.f) {
    LayerState st;
    ring_refresh_lane(lds, r);
    st.pos_cur = ring_acquire<DBG>(lds, r);
    st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    layer8_fetch<DBG, 16, true, SIG_PEND, SIG_OWN, 0, KIND == 0>(bias, bias_pend, wsig, h, st);
    lds_unit<0>(r, st.pos_cur, st.ring[0]);
    lds_unit<1>(r, st.pos_cur, st.ring[1]);
    layer8x_units<DBG, KIND, SIG_PEND, SIG_OWN>(std::make_integer_sequence<int, 64>{}, lds, r, st, bh, bl, mx, acc, bias, bias_pend,
                                                wsig, h, part, k_own, k_pend);
}

// Output layer (2 row blocks, 16 k-steps, one unit per k-step); the lower half of the last hidden layer is
// activated behind its first 15 k-steps (OutPlan).
struct OutState {
    half8 ring[RING_DEPTH][4];
    ActIn in[2][2];
    int pos_cur, pos_nxt;
};

// which half fragment(s) of the last hidden layer's lower half unit U of the output layer activates: fragment 8+k is
// consumed by unit 8+k, so half fragment j (fragment 8 + j/2) has to be finished in a unit < 8 + j/2.  Unit 0 takes
// half fragments 0 and 1, unit u = 1..14 takes half fragment u+1 (deadline 8 + (u+1)/2 > u), unit 15 none: the
// activation VALU work is spread over 15 units instead of packed two-deep into the first 8.
template <int DBG, int U>
struct OutPlan {
    static constexpr bool ACT = U >= 0 && U < 15 && !(DBG & 4);
    static constexpr bool TWO = ACT && U == 0;
    static constexpr int J = U == 0 ? 0 : U + 1;
    static constexpr int T = ACT ? 8 + J / 2 : 8, HS = ACT ? J % 2 : 0;
};

template <int... Us>
constexpr bool out_plan_ok(std::integer_sequence<int, Us...>) {
    int done[16] = {};   // unit in which half fragment j is activated (+1), 0 = never
    bool ok = true;
    auto visit = [&](int U, bool act, bool two, int j) {
        if (!act) return;
        if (done[j]) ok = false;
        done[j] = U + 1;
        if (two) { if (done[j + 1]) ok = false; done[j + 1] = U + 1; }
    };
    (visit(Us, OutPlan<0, Us>::ACT, OutPlan<0, Us>::TWO, OutPlan<0, Us>::J), ...);
    for (int j = 0; j < 16; j++)
        if (!done[j] || done[j] - 1 >= 8 + j / 2) ok = false;   // fragment 8 + j/2 is consumed by unit 8 + j/2
    return ok;
}
static_assert(out_plan_ok(std::make_integer_sequence<int, 16>{}), "OutPlan: every half fragment is ready before its k-step");

template <int DBG, int U>
__device__ __forceinline__ void out_fetch(const float *bias_pend, int h, OutState &st) {
    using P = OutPlan<DBG, U>;
    if constexpr (P::ACT) act_fetch<P::T, P::HS, false>(bias_pend, bias_pend, h, st.in[U & 1][0]);
    if constexpr (P::TWO) act_fetch<P::T, 1, false>(bias_pend, bias_pend, h, st.in[U & 1][1]);
}

template <int DBG, int U>
__device__ __forceinline__ void out_unit(char *lds, Ring &r, OutState &st, half8 (&bh)[16], half8 (&bl)[16],
                                         const f32x16 (&acc)[8], f32x16 (&col)[2], const float *bias_pend, int h, float &part) {
    constexpr int UNITS = 16, RD = RING_DEPTH, UPS = UNITS_PER_SLOT;
    if constexpr (U % UPS == 0 && U != 0) {
        st.pos_cur = ring_acquire<DBG>(lds, r);
        st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    }
    constexpr int UN = U + RD - 1;
    constexpr bool PF = UN < UNITS && !(DBG & 8);
    const int pf_pos = (UN / UPS) == (U / UPS) ? st.pos_cur : st.pos_nxt;
    using P = OutPlan<DBG, U>;
    constexpr bool ACT = P::ACT, TWO = P::TWO;
    constexpr int T = P::T, HS = P::HS;
    ActRegs g0, g1;
    half8(&a)[4] = st.ring[U % RD];
    half8(&nx)[4] = st.ring[UN % RD];
    const ActIn &in0 = st.in[U & 1][0], &in1 = st.in[U & 1][1];
    constexpr bool PF_PREV = U == 0 || ((U - 1 + RD - 1) < UNITS && !(DBG & 8));
    lds_wait<PF_PREV ? 4 : 0>();
    out_fetch<DBG, U + 1>(bias_pend, h, st);
#define SDN_STAGE(K) \
    if constexpr (U % UPS < PIECES / 4 && K < 4 && !(DBG & 1)) ring_issue_piece<4 * (U % UPS) + ((K) & 3)>(lds, r); \
    if constexpr (ACT) act_stage<T, HS, false, K>(acc, in0, bh, bl, part, g0); \
    if constexpr (TWO) act_stage<T, 1, false, K>(acc, in1, bh, bl, part, g1); \
    if constexpr (PF && K < 4) lds_frag<UN % UPS, (K) & 3>(r, pf_pos, nx[(K) & 3]); \
    __builtin_amdgcn_sched_barrier(0);
    col[0] = mfma16(a[0], bh[U], col[0]);
    SDN_STAGE(0)
    col[1] = mfma16(a[2], bh[U], col[1]);
    SDN_STAGE(1)
    col[0] = mfma16(a[1], bh[U], col[0]);
    SDN_STAGE(2)
    col[1] = mfma16(a[3], bh[U], col[1]);
    SDN_STAGE(3)
    col[0] = mfma16(a[0], bl[U], col[0]);
    SDN_STAGE(4)
    col[1] = mfma16(a[2], bl[U], col[1]);
    SDN_STAGE(5)
#undef SDN_STAGE
}

template <int DBG, int... Us>
__device__ __forceinline__ void out_units(std::integer_sequence<int, Us...>, char *lds, Ring &r, OutState &st,
                                          half8 (&bh)[16], half8 (&bl)[16], const f32x16 (&acc)[8], f32x16 (&col)[2],
                                          const float *bias_pend, int h, float &part) {
    (out_unit<DBG, Us>(lds, r, st, bh, bl, acc, col, bias_pend, h, part), ...);
}

template <int DBG>
__device__ __forceinline__ void layer_out(char *lds, Ring &r, half8 (&bh)[16], half8 (&bl)[16], const f32x16 (&acc)[8],
                                          f32x16 (&col)[2], const float *bias_pend, int h, float &part) {
    OutState st;
    ring_refresh_lane(lds, r);
    st.pos_cur = ring_acquire<DBG>(lds, r);
    st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    out_fetch<DBG, 0>(bias_pend, h, st);
    lds_unit<0>(r, st.pos_cur, st.ring[0]);
    lds_unit<1>(r, st.pos_cur, st.ring[1]);
    out_units<DBG>(std::make_integer_sequence<int, 16>{}, lds, r, st, bh, bl, acc, col, bias_pend, h, part);
}

// DBG & 512 (timing experiment, ablation builds): cycles of workgroup-thread 0 per segment of a pass, summed in LDS --
// 0 inputs (encode stage / staging), 1 fc_1, 2..6 fc_2..fc_6, 7 fc_out_c, 8 volume rendering, 9 everything between passes of
// different groups; 10 = passes; 11 = the colour-skip decision (early sigma + ballot), 12 = passes whose colour branch was skipped.  s_memtime is an SMEM operation: the compiler waits lgkmcnt(0) for it, which is only stricter
// than the hand-counted LDS waits around it (segment boundaries have no fragment reads in flight).
template <int DBG>
__device__ __forceinline__ void seg_tick(char *lds, int idx, unsigned &tprev) {
    if constexpr (DBG & 512) {
        const unsigned now = (unsigned)__builtin_readcyclecounter();
        if (threadIdx.x == 0) reinterpret_cast<unsigned *>(lds + LDS_TIMERS)[idx] += now - tprev;
        tprev = now;
    }
}

// CT = number of split terms of the colour layers fc_5 / fc_6 (3, or 2 = without the Whi.Xlo products)
// FUSED = the encode stage runs inside this kernel (field_kernel): a pass's B fragments, distances and labels come from
//         enc_place / enc_level instead of the feature buffer, the ray flags from the intersections themselves
template <int DBG, int CT, int MODE = MODE_BUFFER>
__global__ __launch_bounds__(256, 1) void mlp_kernel(const MlpParams p) {
    constexpr bool FUSED = MODE == MODE_FUSED || MODE == MODE_FUSED_AUX, AUX = MODE == MODE_FUSED_AUX, RAW = MODE == MODE_RAW;
    // Colour-branch skipping (field_kernel): a sample with relu(sigma) * dist == 0 has volume-rendering weight EXACTLY 0
    // (mc_utils.py:154-161: weights = (1 - exp(-relu(sigma) * dists)) * T), so its colour is multiplied by zero; when that holds
    // for all 128 samples of a workgroup's pass, fc_5 / fc_6 / fc_out_c (35 % of the pass's matrix instructions, 18 of its 46
    // ring slots) are not evaluated at all -- net_out is bit-identical.  sigma is complete only after fc_4's lower half has
    // been activated, which normally happens as fc_5's pending work: sigma_lower_half computes that half's contribution
    // ahead on a copy of the running sum (a few % of a pass), the decision is a wave ballot combined over the 4 waves (they
    // share the weight ring).  (Taking the decision inside fc_5, behind its pending work, costs nothing extra per pass but saves
    // less per skipped pass -- measured on the same box: -4.8 % vs -6.9 % of the kernel time.)
    constexpr bool SKIP = MODE == MODE_FUSED;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_TOTAL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int q = j & 3;

    float *cst = reinterpret_cast<float *>(lds + LDS_CONST);
    // the frame mean of the sky features arrives straight from sky_kernel (no host-side copy into the constant block).
    // ONE writer per LDS word: two waves writing the same word without a barrier in between land in either order.
    static_assert(C_SKY_AVG + OUTC == C_TOTAL, "sky_avg is the tail of the constant block");
    for (int i = threadIdx.x; i < C_TOTAL; i += 256)
        cst[i] = (p.sky_avg && i >= C_SKY_AVG) ? p.sky_avg[i - C_SKY_AVG] : p.consts[i];
    // FUSED: the encode stage reads its tables from LDS (a copy of the parameter block with the three pointers redirected)
    EncParams enc = p.enc;
    if constexpr (FUSED) {
        float *e_scales = reinterpret_cast<float *>(lds + LDS_ENC_SCALES), *e_lin = reinterpret_cast<float *>(lds + LDS_ENC_LIN);
        uint8_t *e_lut = reinterpret_cast<uint8_t *>(lds + LDS_ENC_LUT);
        if (threadIdx.x < NLEV) e_scales[threadIdx.x] = p.enc.scales[threadIdx.x];
        if (threadIdx.x < p.enc.ns + 1) e_lin[threadIdx.x] = p.enc.lin[threadIdx.x];
        for (int i = threadIdx.x; i < 1024; i += 256) e_lut[i] = p.enc.lut[i];
        enc.scales = e_scales; enc.lin = e_lin; enc.lut = e_lut;
        if (p.cam_ori_dev) {   // (uniform: three scalar loads)
            enc.ori[0] = p.cam_ori_dev[0]; enc.ori[1] = p.cam_ori_dev[1]; enc.ori[2] = p.cam_ori_dev[2];
        }
    }
    __syncthreads();

    Ring r;
    r.slots_per_pass = SLOTS_PER_PASS;
    r.wbytes = reinterpret_cast<const char *>(p.wpk);
    r.g = 0;
    r.wave = __builtin_amdgcn_readfirstlane(wave);
    r.lane = lane;
    r.voff = r.wave * (PIECES * 1024) + lane * 16;
    r.lds_lane = (unsigned)(size_t)(const lds_char *)(lds + LDS_RING) + lane * 16;
    r.src_delta = r.wave * (PIECES * 1024) - (int)(unsigned)(size_t)(const lds_char *)(lds + LDS_RING);
#pragma unroll
    for (int sl = 0; sl < DMA_AHEAD; sl++) ring_issue(lds, r, sl, sl);
    r.next_in_pass = DMA_AHEAD;

    // Input prefetch: the next pass's 16 KiB of features (+ label, dist) are loaded into otherwise idle AGPRs while the
    // output layer of the current pass runs, so the HBM latency of the pass-start loads (3.5 % of a pass) is hidden.
    // The loads are inline asm (the compiler would place its own, draining waits) and are consumed only through
    // v_accvgpr_read asm behind a hand-counted wait: after them the output layer always issues its 4 slots x 4 ring
    // DMAs, so "vmcnt(16)" at the next pass start means the prefetch has landed (more vm operations in between -- the
    // stores at a group's end -- only make the wait stricter).
    // The landing registers are the PHYSICAL AGPRs a[190:255], named in the asm text: as C++ values they would be
    // loop-carried, and hipcc keeps loop-carried values in VGPRs, i.e. copies them out of the AGPRs right behind the
    // load -- before the data has landed.  The kernel's own allocation stays below a190 (tools/check_lds_hazards.py
    // verifies that no other instruction touches a[190:255]).
    unsigned t_seg = 0;
    if constexpr (DBG & 512) {
        if (threadIdx.x < 16) reinterpret_cast<unsigned *>(lds + LDS_TIMERS)[threadIdx.x] = 0u;
        t_seg = (unsigned)__builtin_readcyclecounter();
    }
    long pf_tc = -1;            // pass whose inputs sit in a[190:255] (wave-uniform), -1: none
    unsigned long long t_stage = 0, t_kernel0 = 0;
    unsigned n_pass = 0;
    if constexpr (DBG & 128) t_kernel0 = __builtin_readcyclecounter();
    const int n_groups = (p.n_tiles + 3) >> 2;
    // Group schedule.  Static: workgroup b takes groups b, b + G, b + 2G, ... (G = gridDim.x).  With a ticket counter the
    // first TWO rounds are static and every later group is drawn from the counter one group AHEAD of its use (the draw
    // of group n+2 is issued at the start of group n, so the group after the current one is always known: its first pass
    // is prefetched during the current group's last).  Groups are independent, so net_out does not depend on the schedule.
    // (The atomic's round trip IS waited for at the group's start: hipcc consumes an atomic's result at once -- `vmcnt(0)`
    // behind it, once per group, <= 0.6 % of a group's time.  Keeping the draw pending for one more group was tried in
    // round 6: the result is then a loop-carried value, which hipcc copies into its carrier register right behind the
    // atomic, with the same wait; returning it into a reserved physical register is what the AGPR prefetch of the
    // two-kernel form does, and is not worth a second such contract here.)
    // The workgroup's decision words (LDS_FLAGS), accessed with explicit ds instructions: as a `volatile int *` they became FLAT
    // loads / stores (sc0 sc1) whose 64-bit addresses hipcc kept in scratch memory and each of which it followed with
    // s_waitcnt vmcnt(0) -- a complete drain of the weight ring's DMAs at every group start, after every pass (termination
    // ballot) and in every colour-skip decision (seen in the ISA of rounds 4-5: 17 flat operations, 48 of the kernel's 100 B of scratch).
#define flags_a lds_addr(lds + LDS_FLAGS)
    auto flag_put = [&](int word, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(flags_a + 4u * (unsigned)word), "v"(v) : "memory"); };
    auto flag_get4 = [&](int word0) -> i32x4v {     // words word0 .. word0 + 3 (16-byte aligned), landed
        i32x4v f;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(flags_a + 4u * (unsigned)word0) : "memory");
        return f;
    };
    int grp = blockIdx.x, grp_next = blockIdx.x + (int)gridDim.x;
    // FUSED: the first intersection of this lane's ray in group g (0 = none / no ray).  The next group's is loaded at the
    // START of the current one, so a group does not begin with an exposed round trip to memory
    auto first_vox = [&](int g) -> int {
        const int t = g * 4 + wave, ry = t * RAYS_PER_TILE + (j >> 2);
        if (!(g < n_groups && t < p.n_tiles && ry < p.R && enc.win.valid(ry))) return 0;
        return enc.voxel_id[(size_t)enc.win.src(ry) * enc.M];
    };
    int vox_cur = 0;
    if constexpr (FUSED) vox_cur = first_vox(grp);
    int cold = 0;     // (uniform) consecutive groups of this workgroup that were found dense, or not asked (colour-branch skipping)
    while (grp < n_groups) {
        const int tile = grp * 4 + wave;
        const bool tile_ok = tile < p.n_tiles;
        const int tile_s = grp * 4 + r.wave;              // the same as scalars (r.wave went through readfirstlane)
        const bool tile_ok_s = tile_s < p.n_tiles;
        const int ray = tile * RAYS_PER_TILE + (j >> 2);
        const bool ray_ok = tile_ok && ray < p.R && (!FUSED || p.win.valid(ray));
        const int rl = ray_ok ? ray : p.R - 1;            // FUSED: local ray / the same ray in the frame-wide arrays
        const int rr = FUSED ? enc.win.src(rl) : 0;
        uint8_t flag;                                      // bit 0 sky_only, bit 1 nosky (FUSED: bit 1 is known at the group's end)
        int vox_nxt = 0;
        if constexpr (FUSED) {
            flag = vox_cur != 0 ? (uint8_t)0 : (uint8_t)1;   // scenedreamer.py:337
            vox_nxt = first_vox(grp_next);
        } else if constexpr (RAW) flag = tile_ok ? (uint8_t)0 : (uint8_t)1;   // every tile holds at least one row
        else flag = ray_ok ? p.rayflag[ray] : (uint8_t)1;
        bool gnd = false;                                  // FUSED: any sample of the ray at world x <= 1 (:380)
        int drawn = grp_next + (int)gridDim.x;            // the group after next: static stride, or ...
        if (p.ticket && threadIdx.x == 0) drawn = 2 * (int)gridDim.x + atomicAdd(p.ticket, 1);   // ... the next undrawn one
        // (AUX also returns the per-sample sigma / colour of rays that hit nothing -- the reference evaluates them -- so it skips no group)
        const bool any_hit = AUX ? tile_ok : __any(!(flag & 1));
        // workgroup-uniform decisions: skip the group when none of its 32 rays hits anything; everybody learns the draw
        if (lane == 0) flag_put(wave, any_hit ? 1 : 0);
        if (threadIdx.x == 0) flag_put(4, drawn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // readfirstlane makes the decisions provably uniform: otherwise every loop-carried ring counter / pointer is
        // classified divergent, lives in VGPRs (spills!) and the DMA cannot use scalar addressing
        const i32x4v fh = flag_get4(0), fd = flag_get4(4);
        const bool grp_hit = __builtin_amdgcn_readfirstlane(fh[0] | fh[1] | fh[2] | fh[3]) != 0;
        const int grp_next2 = __builtin_amdgcn_readfirstlane(fd[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the next group rewrites the flags only after everyone has read them

        float outq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float carry = 0.f, tsum = 0.f;
        int n_done = 0, n_colour = 0;
        // (uniform) whether the next pass of this group takes the colour-skip decision at all.  Across groups: after 6 groups in a row
        // whose first pass was less than 1/4 empty (a field with surfaces: rays saturate, nothing is empty) only every 4th group
        // still decides -- there the early sigma was +1 % of the frame for nothing (tools/bench_opaque.py); on the benchmark
        // frames the rule costs about one point of skipped passes (dense stretches end somewhere)
        bool skip_test = SKIP && !p.no_colour_skip && (cold < 6 || (cold & 3) == 0);
        if (SKIP && !skip_test) cold++;

        for (int ch = 0; grp_hit && ch < p.nch; ch++) {
            const size_t tc = (size_t)(tile_ok ? tile : 0) * p.nch + ch;
            half8 bh[16], bl[16];
            f32x16 acc[8];
            MxState mx;
            if constexpr (CT == 6) {
#pragma unroll
                for (int k = 0; k < 4; k++) mx.bm[k] = 0.f;
            }
            seg_tick<DBG>(lds, 9, t_seg);
            const float *fin = p.feat + (tc * 8 * 64 + lane) * 8;
            unsigned long long t_in0 = 0;
            if constexpr (DBG & 128) t_in0 = __builtin_readcyclecounter();
            const long tc_s = (long)(tile_ok_s ? tile_s : 0) * p.nch + ch;    // tc as a scalar
            int lab;
            float dist;
            float smp_depth = 0.f;   // AUX: this lane's sample depth
            float raw[8][8];
            // RAW: this lane's row of the [R, 128] feature matrix (clamped: lanes past the end evaluate the last row, store nothing)
            const long row = (long)(tile_ok ? tile : 0) * 256 + ch * 32 + j;
            if constexpr (RAW) {
                const long rc = row < p.R ? row : (long)p.R - 1;
                lab = p.label[rc];
                dist = 0.f;
                const float *src = p.feat + rc * FEAT + 8 * h;   // kmap_first: k-step s, lane half h = features 16 s + 8 h .. + 7
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const float4 a = *reinterpret_cast<const float4 *>(src + 16 * s), b = *reinterpret_cast<const float4 *>(src + 16 * s + 4);
                    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                    split8(v, bh[s], bl[s]);
                }
            } else if constexpr (FUSED) {
                // ---- the encode stage of THIS pass, by this wave, into its own B-fragment registers (the accumulators, the
                //      fragment ring and the fp6 state are dead here, so the gathers of several levels can be in flight).
                //      The loads are ordinary ones: hipcc's own vmcnt waits also retire the ring DMAs issued before them
                //      (vector memory completes in order) -- a stricter wait than the ring's counted ones, never a wrong one.
                RayBoxes rb;
                float dd[3];
                enc_load_ray(enc, rr, rb, dd);
                const EncSample es = enc_place(enc, rb, dd, rl, ch * SAMP_PER_STEP + (j & 3), ray_ok);
                gnd = gnd || es.gnd;
                lab = es.label;
                dist = es.dist;
                if constexpr (AUX) smp_depth = es.depth;
                const bool use_feat = AUX ? ray_ok : !(flag & 1);
                // 4 levels' gathers (64 x 16 B per lane) in flight at a time: two round trips per pass instead of eight
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    float res[4][8];
                    enc_levels<4>(enc, es, 4 * b, h, use_feat, res);
#pragma unroll
                    for (int t = 0; t < 4; t++) split8(res[t], bh[4 * b + t], bl[4 * b + t]);
                }
            } else {
            if (pf_tc == tc_s && !(DBG & 256)) {
                if constexpr (DBG & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#define SDN_PF_READ8(S, A0, A1, A2, A3, A4, A5, A6, A7) \
    asm volatile("v_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 "\n\tv_accvgpr_read_b32 %2, a" #A2 \
                 "\n\tv_accvgpr_read_b32 %3, a" #A3 "\n\tv_accvgpr_read_b32 %4, a" #A4 "\n\tv_accvgpr_read_b32 %5, a" #A5 \
                 "\n\tv_accvgpr_read_b32 %6, a" #A6 "\n\tv_accvgpr_read_b32 %7, a" #A7 \
                 : "=v"(raw[S][0]), "=v"(raw[S][1]), "=v"(raw[S][2]), "=v"(raw[S][3]), "=v"(raw[S][4]), "=v"(raw[S][5]), \
                   "=v"(raw[S][6]), "=v"(raw[S][7]))
                SDN_PF_READ8(0, 190, 191, 192, 193, 194, 195, 196, 197);
                SDN_PF_READ8(1, 198, 199, 200, 201, 202, 203, 204, 205);
                SDN_PF_READ8(2, 206, 207, 208, 209, 210, 211, 212, 213);
                SDN_PF_READ8(3, 214, 215, 216, 217, 218, 219, 220, 221);
                SDN_PF_READ8(4, 222, 223, 224, 225, 226, 227, 228, 229);
                SDN_PF_READ8(5, 230, 231, 232, 233, 234, 235, 236, 237);
                SDN_PF_READ8(6, 238, 239, 240, 241, 242, 243, 244, 245);
                SDN_PF_READ8(7, 246, 247, 248, 249, 250, 251, 252, 253);
#undef SDN_PF_READ8
                asm volatile("v_accvgpr_read_b32 %0, a254\n\tv_accvgpr_read_b32 %1, a255" : "=v"(lab), "=v"(dist));
                if (!tile_ok) dist = 0.f;
            } else {
                // first pass of the kernel / after a skipped group: load here.  Inline asm as well, so that hipcc's wait
                // insertion sees no vector-memory loads at all in this loop (with ordinary loads on this path it puts
                // vmcnt(0) in front of the AGPR reads of the other path: a full drain of the weight ring per pass)
                f32x4 t[16];
                const char *base = reinterpret_cast<const char *>(fin);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const char *a = base + k * 4096;
                    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                                 "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:2064"
                                 : "=&v"(t[4 * k]), "=&v"(t[4 * k + 1]), "=&v"(t[4 * k + 2]), "=&v"(t[4 * k + 3]) : "v"(a) : "memory");
                }
                asm volatile("global_load_ubyte %0, %2, off\n\tglobal_load_dword %1, %3, off"
                             : "=&v"(lab), "=&v"(dist) : "v"(p.label + tc * 32 + j), "v"(p.dist + tc * 32 + j) : "memory");
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                               "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]),
                               "+v"(lab), "+v"(dist)
                             :: "memory");
                if (!tile_ok) dist = 0.f;
#pragma unroll
                for (int s = 0; s < 8; s++)
#pragma unroll
                    for (int e = 0; e < 8; e++) raw[s][e] = t[2 * s + (e >> 2)][e & 3];
            }
#pragma unroll
            for (int s = 0; s < 8; s++) {   // encode_kernel wrote the fragment already split: dwords 0..3 = f16 hi, 4..7 = f16 lo
                const u32x4v hw = {__builtin_bit_cast(unsigned int, raw[s][0]), __builtin_bit_cast(unsigned int, raw[s][1]),
                                   __builtin_bit_cast(unsigned int, raw[s][2]), __builtin_bit_cast(unsigned int, raw[s][3])};
                const u32x4v lw = {__builtin_bit_cast(unsigned int, raw[s][4]), __builtin_bit_cast(unsigned int, raw[s][5]),
                                   __builtin_bit_cast(unsigned int, raw[s][6]), __builtin_bit_cast(unsigned int, raw[s][7])};
                bh[s] = __builtin_bit_cast(half8, hw);
                bl[s] = __builtin_bit_cast(half8, lw);
            }
            }   // !FUSED
            if constexpr (DBG & 128) {
                asm volatile("s_waitcnt vmcnt(0)" ::"v"(bh[7]), "v"(bl[7]) : "memory");
                t_stage += __builtin_readcyclecounter() - t_in0;
                n_pass++;
            }
            if constexpr (DBG & 512) asm volatile("s_waitcnt vmcnt(0)" ::"v"(bh[7]), "v"(bl[7]) : "memory");
            seg_tick<DBG>(lds, 0, t_seg);
            float part = 0.f;
            bool colour_skipped = false;
            const float *wsig = cst + C_WSIGMA;
            // ---- fc_1: 8 k-steps; fragments 0..6 of its upper half are activated behind its own lower half, fragment 7
            //      right after it, its lower half behind fc_2's head ---------------------------------------------------
            const float *bias1 = cst + C_LABEL_BIAS + lab * HID;
            layer8<DBG, 8, false, false, false>(lds, r, bh, bl, acc, bias1, bias1, wsig, h, part, TRUNK_K, TRUNK_K);
            act_step<7, false>(acc, bias1, wsig, h, bh, bl, part, TRUNK_K);   // fragments 0..6 were activated inside the layer
            seg_tick<DBG>(lds, 1, t_seg);
            // ---- fc_2 .. fc_6.  fc_4 (l == 2) feeds the density head (layers.py:114): its upper half is activated
            //      inside l == 2, its lower half as the pending work of l == 3 ----------------------------------------
#pragma unroll 1
            for (int l = 0; l < 5; l++) {   // (straight-line code instead of this loop: 1.5 KB of scratch spills -- tried)
                const float *bias = cst + C_BETA + l * HID;
                const float *bias_pend = l == 0 ? bias1 : bias - HID;   // the previous layer's (its lower half is pending)
                // the trunk's packed weights carry 2^TRUNK_SHIFT (pack_kernel): its accumulators are descaled in the bias fma.
                // (this layer is fc_(l+2), the pending one fc_(l+1); literals per branch -- as run-time scalars they cost spills)
                constexpr float tk = TRUNK_K;
                if constexpr (CT == 6) {   // colour layers: f16 Whi.Xhi + fp6 corrections (layer8x)
                    if (l == 2) layer8x<DBG, 0, false, true>(lds, r, bh, bl, mx, acc, bias, bias_pend, wsig, h, part, tk, tk);
                    else if (l == 3) layer8x<DBG, 1, true, false>(lds, r, bh, bl, mx, acc, bias, bias_pend, wsig, h, part, 1.f, tk);
                    else if (l == 4) layer8x<DBG, 2, false, false>(lds, r, bh, bl, mx, acc, bias, bias_pend, wsig, h, part, 1.f, 1.f);
                    else layer8<DBG, 16, true, false, false>(lds, r, bh, bl, acc, bias, bias_pend, wsig, h, part, tk, tk);
                } else {
                    // fc_4's upper half feeds fc_5, fc_5's activations feed fc_5 / fc_6: no lo parts when those are 2-term
                    if (l == 2) layer8<DBG, 16, true, false, true, 3, true, CT == 3>(lds, r, bh, bl, acc, bias, bias_pend, wsig, h, part, tk, tk);
                    else if (l == 3) layer8<DBG, 16, true, true, false, CT, CT == 3, CT == 3>(lds, r, bh, bl, acc, bias, bias_pend, wsig, h, part, 1.f, tk);
                    else if (CT != 3 && l == 4) layer8<DBG, 16, true, false, false, CT, CT == 3, true>(lds, r, bh, bl, acc, bias, bias_pend, wsig, h, part, 1.f, 1.f);
                    else if (l == 4) layer8<DBG, 16, true, false, false>(lds, r, bh, bl, acc, bias, bias_pend, wsig, h, part, 1.f, 1.f);
                    else layer8<DBG, 16, true, false, false>(lds, r, bh, bl, acc, bias, bias_pend, wsig, h, part, tk, tk);
                }
                seg_tick<DBG>(lds, 2 + l, t_seg);
                if constexpr (SKIP) {
                    if (l == 2 && skip_test) {
                        // sigma of this pass: `part` holds fc_4's upper half, the lower half's terms are added on a copy (fc_5's
                        // pending work adds them to `part` itself, in the same order, if the pass goes on)
                        float part_e = part;
                        sigma_lower_half(std::make_integer_sequence<int, 16>{}, acc, bias, wsig, h, bh, bl, part_e, tk);
                        const float sigma_e = part_e + __shfl_xor(part_e, 32) + cst[C_BSIGMA];
                        const bool zero_w = !ray_ok || (flag & 1) || fmaxf(sigma_e, 0.f) * dist == 0.f;
                        const int wave_zero = __popcll(__ballot(zero_w));          // lanes: 2 per sample
                        // ONE barrier: flags[8..11] are written only here, and a wave reaches its next write only through the ring
                        // barriers of at least one whole layer, which nobody passes before having read these
                        if (lane == 0) flag_put(8 + wave, wave_zero);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        const i32x4v fz = flag_get4(8);
                        const int grp_zero = __builtin_amdgcn_readfirstlane(fz[0] + fz[1] + fz[2] + fz[3]);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        seg_tick<DBG>(lds, 11, t_seg);
                        // Whether a pass is empty hardly depends on the depth along the rays (measured on the benchmark frames,
                        // tools/dbg_sigma_stats.py: P(next pass of the group empty | this one empty) = 0.89 - 0.93, | this one less
                        // than 3/4 empty: 0.000 - 0.002), so a group stops taking the decision -- the work of computing sigma
                        // ahead -- after a pass that was not at least 3/4 empty.  Purely a cost heuristic: net_out cannot change.
                        // (Tried on top: a WAVE whose own 32 samples are empty sitting the colour layers out while the others work --
                        //  barriers and its share of the ring DMA only.  Same box: 15.03 -> 15.55 ms, 18.49 -> 19.08 ms: the idle
                        //  SIMD buys the other three nothing, the extra decisions cost.  Not kept.)
                        skip_test = grp_zero >= 192;
                        if (ch == 0) cold = grp_zero < 64 ? cold + 1 : 0;
                        if (grp_zero == 256) { colour_skipped = true; break; }
                    }
                }
            }
            if constexpr (SKIP) {
                if (colour_skipped) {
                    // every weight of the pass is exactly zero: carry, tsum and outq would all be incremented by +0
                    ring_restart(lds, r);
                    if constexpr (DBG & 512) {
                        if (threadIdx.x == 0) {
                            reinterpret_cast<unsigned *>(lds + LDS_TIMERS)[10] += 1u;
                            reinterpret_cast<unsigned *>(lds + LDS_TIMERS)[12] += 1u;
                        }
                    }
                    n_done = ch + 1;
                    continue;   // (no termination ballot: the transmittances did not change since the last one)
                }
            }
            n_colour++;
            // ---- fc_out_c ------------------------------------------------------------------------------------------
            {   // inputs of the next pass of this wave: the next step of this tile, or the first step of its next group
                long tn = tc_s + 1;
                bool has_next = tile_ok_s;
                if (ch + 1 == p.nch) {
                    // first step of this wave's tile in the workgroup's next group (wasted if that group hits nothing)
                    const int tile2 = grp_next * 4 + r.wave;
                    has_next = tile2 < p.n_tiles;
                    tn = (long)tile2 * p.nch;
                }
                pf_tc = -1;
                if (MODE == MODE_BUFFER && has_next && !(DBG & 256)) {
                    pf_tc = tn;
                    const char *base = reinterpret_cast<const char *>(p.feat + ((size_t)tn * 8 * 64 + lane) * 8);
                    // k-steps 2k, 2k+1 (2048 B apart), two 16-B halves each -> a[190+16k : 205+16k]
#define SDN_PF_LOAD4(K, R0, R1, R2, R3) \
    asm volatile("global_load_dwordx4 a[" #R0 ":" #R0 "+3], %0, off\n\tglobal_load_dwordx4 a[" #R1 ":" #R1 "+3], %0, off offset:16\n\t" \
                 "global_load_dwordx4 a[" #R2 ":" #R2 "+3], %0, off offset:2048\n\tglobal_load_dwordx4 a[" #R3 ":" #R3 "+3], %0, off offset:2064" \
                 ::"v"(base + (K) * 4096) : "memory", SDN_PF_CLOBBERS)
                    // (the clobber list is what makes the kernel's register count include a[190:255])
#define SDN_PF_CLOBBERS "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253"
                    SDN_PF_LOAD4(0, 190, 194, 198, 202);
                    SDN_PF_LOAD4(1, 206, 210, 214, 218);
                    SDN_PF_LOAD4(2, 222, 226, 230, 234);
                    SDN_PF_LOAD4(3, 238, 242, 246, 250);
#undef SDN_PF_CLOBBERS
#undef SDN_PF_LOAD4
                    asm volatile("global_load_ubyte a254, %0, off" ::"v"(p.label + (size_t)tn * 32 + j) : "memory", "a254");
                    asm volatile("global_load_dword a255, %0, off" ::"v"(p.dist + (size_t)tn * 32 + j) : "memory", "a255");
                }
            }
            f32x16 col[2];
            col[0] = bias_block<0>(cst + C_BC, h);
            col[1] = bias_block<1>(cst + C_BC, h);
            layer_out<DBG>(lds, r, bh, bl, acc, col, cst + C_BETA + 4 * HID, h, part);
            seg_tick<DBG>(lds, 7, t_seg);
            const float sigma = part + __shfl_xor(part, 32) + cst[C_BSIGMA];
            if constexpr (RAW) {   // LightningMLP.forward's outputs for this lane's row: (sigma, c), layers.py:114, :124
                if (tile_ok && row < p.R) {
                    if (h == 0) p.sigma_out[row] = sigma;
#pragma unroll
                    for (int ib = 0; ib < 2; ib++)
#pragma unroll
                        for (int g4 = 0; g4 < 4; g4++)   // registers 4 g4 .. 4 g4 + 3 of row block ib = features 32 ib + 8 g4 + 4 h + e
                            *reinterpret_cast<float4 *>(p.net_out + (size_t)row * OUTC + 32 * ib + 8 * g4 + 4 * h) =
                                make_float4(col[ib][4 * g4], col[ib][4 * g4 + 1], col[ib][4 * g4 + 2], col[ib][4 * g4 + 3]);
                }
                n_done = ch + 1;
                continue;
            }
            // ---- volume rendering (mc_utils.py:154-161) over the 4 samples of each ray in this pass ---------------
            const float fe = fmaxf(sigma, 0.f) * dist;
            float incl = fe;
            float up = quad_dpp<QUAD_UP1>(incl);
            if (q >= 1) incl += up;
            up = quad_dpp<QUAD_UP2>(incl);
            if (q >= 2) incl += up;
            float ex = quad_dpp<QUAD_UP1>(incl);
            if (q == 0) ex = 0.f;
            const float excl = carry + ex;
            const float wgt = (1.f - __expf(-fe)) * __expf(-excl);
            carry += quad_dpp<QUAD_LAST>(incl);
            tsum += wgt;
            if constexpr (AUX) {   // the per-sample return values of Generator._forward_perpix
                const int sidx = ch * SAMP_PER_STEP + q;
                if (ray_ok && sidx < p.ns) {
                    const size_t smp = (size_t)p.win.out_row(ray) * p.ns + sidx;
                    if (h == 0) {
                        if (p.w_out) p.w_out[smp] = (flag & 1) ? 0.f : wgt;
                        if (p.depth_out) p.depth_out[smp] = smp_depth;
                        if (p.sig_out) p.sig_out[smp] = sigma;
                    }
                    if (p.col_out) {
#pragma unroll
                        for (int ib = 0; ib < 2; ib++)
#pragma unroll
                            for (int g4 = 0; g4 < 4; g4++)   // registers 4 g4 .. 4 g4 + 3 of row block ib = features 32 ib + 8 g4 + 4 h + e
                                *reinterpret_cast<float4 *>(p.col_out + smp * OUTC + 32 * ib + 8 * g4 + 4 * h) =
                                    make_float4(col[ib][4 * g4], col[ib][4 * g4 + 1], col[ib][4 * g4 + 2], col[ib][4 * g4 + 3]);
                    }
                }
            }
#pragma unroll
            for (int ib = 0; ib < 2; ib++)
#pragma unroll
                for (int rr = 0; rr < 16; rr++) {
                    const float rgb = fminf(fmaxf(col[ib][rr], -1.f), 1.f) + 1.f;  // scenedreamer.py:408
                    float v = wgt * rgb;
                    v += quad_dpp<QUAD_XOR1>(v);   // sum over the 4 samples of the ray held by this quad
                    v += quad_dpp<QUAD_XOR2>(v);
                    if ((rr >> 2) == q) outq[ib][rr & 3] += v;
                }
            // ---- early ray termination (north star: wavefront ballots): once the transmittance exp(-carry) of EVERY ray of
            //      the workgroup's 32 is below eps, the remaining samples can change net_out by at most 2 eps (their weights
            //      sum to < eps and that mass goes to the sky term instead): skip the group's remaining passes.  The
            //      decision is a wave ballot combined over the 4 waves, because they share the weight ring / barriers.
            if constexpr (DBG & 512) {
                asm volatile("" ::"v"(outq[0][0]), "v"(outq[1][3]), "v"(carry), "v"(tsum));
                seg_tick<DBG>(lds, 8, t_seg);
                if (threadIdx.x == 0) reinterpret_cast<unsigned *>(lds + LDS_TIMERS)[10] += 1u;
            }
            n_done = ch + 1;
            if (p.term_depth > 0.f && ch + 1 < p.nch) {
                const bool opaque = !ray_ok || (flag & 1) || carry > p.term_depth;
                const bool wave_done = __all(opaque);
                if (lane == 0) flag_put(wave, wave_done ? 1 : 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const i32x4v ft = flag_get4(0);
                const bool grp_done = __builtin_amdgcn_readfirstlane(ft[0] & ft[1] & ft[2] & ft[3]) != 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (grp_done) break;
            }
        }
#ifndef SDN_NO_TERM_GND   // (A/B switch for timing the main loop without this cold block; never set for a product build)
        if constexpr (FUSED) {
            // early termination dropped the group's remaining passes: their samples still take part in `is_gnd` (any sample of
            // the ray at world x <= 1, scenedreamer.py:380-382), as they do in encode_kernel -- place them (no gathers, no MLP)
            if (grp_hit && n_done < p.nch) {
                RayBoxes rb;
                float dd[3];
                enc_load_ray(enc, rr, rb, dd);
#pragma unroll 1
                for (int c2 = n_done; c2 < p.nch; c2++)
                    gnd = gnd || enc_place(enc, rb, dd, rl, c2 * SAMP_PER_STEP + (j & 3), ray_ok).gnd;
            }
        }
#endif
        if (p.passes && threadIdx.x == 0) p.passes[grp] = (uint8_t)n_done;   // passes this group went through (tests / bench)
        if (p.colour_passes && threadIdx.x == 0) p.colour_passes[grp] = (uint8_t)n_colour;   // ... and how many of them ran the colour branch

        // ---- blend the sky, store ---------------------------------------------------------------------------------
        if constexpr (!RAW) {
        tsum += quad_dpp<QUAD_XOR1>(tsum);
        tsum += quad_dpp<QUAD_XOR2>(tsum);
        if constexpr (FUSED) {   // nosky = the ray's last intersection is a voxel, or one of its samples lies at world x <= 1 (:335, :382)
            int g = (int)gnd;
            g |= quad_dpp<QUAD_XOR1>(g);
            g |= quad_dpp<QUAD_XOR2>(g);
            const bool last_hit = ray_ok && enc.voxel_id[(size_t)rr * enc.M + (enc.M - 1)] != 0;
            if (last_hit || g) flag |= 2;
        }
        const bool sky_only = flag & 1, nosky = flag & 2;
        if (sky_only) tsum = 0.f;  // scenedreamer.py:376
        const float sky_w = 1.f - tsum;
        if (ray_ok) {
#pragma unroll
            for (int ib = 0; ib < 2; ib++) {
                const int f0 = 32 * ib + 8 * q + 4 * h;   // this lane owns features f0 .. f0+3 of its ray
                const float4 sc = *reinterpret_cast<const float4 *>(p.sky_c + (size_t)p.win.src(ray) * OUTC + f0);
                const float4 sa = *reinterpret_cast<const float4 *>(cst + C_SKY_AVG + f0);
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, sav[4] = {sa.x, sa.y, sa.z, sa.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float sky = nosky ? (scv[e] * 0.f + sav[e]) : scv[e];          // :401, mask in {0,1}
                    const float rgb_sky = fminf(fmaxf(sky, -1.f), 1.f) + 1.f;
                    o[e] = (sky_only ? 0.f : outq[ib][e]) + sky_w * rgb_sky - 1.f;       // :410-413
                    if constexpr (AUX) {
                        if (p.skyb_out) p.skyb_out[(size_t)p.win.out_row(ray) * OUTC + f0 + e] = sky;
                    }
                }
                if constexpr (AUX) {
                    if (p.nosky_out && ib == 0 && q == 0 && h == 0) p.nosky_out[p.win.out_row(ray)] = nosky ? 1 : 0;
                }
                *reinterpret_cast<float4 *>(p.net_out + (size_t)p.win.out_row(ray) * OUTC + f0) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        }   // !RAW
        grp = grp_next;
        grp_next = grp_next2;
        vox_cur = vox_nxt;
    }
    // the ring runs DMA_AHEAD slots ahead of the last pass: let it land before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the last workgroup to leave resets the ticket counter for the next launch
    if (p.ticket && threadIdx.x == 0 && atomicAdd(p.ticket + 1, 1) == (int)gridDim.x - 1) {
        p.ticket[0] = 0;
        p.ticket[1] = 0;
    }
    if constexpr (DBG & 512) {   // per-segment cycles of this workgroup's thread 0 into its first net_out row (floats 3..13)
        __syncthreads();
        if (threadIdx.x < 13) p.net_out[(size_t)(blockIdx.x * 4) * OUTC + 3 + threadIdx.x] = (float)reinterpret_cast<unsigned *>(lds + LDS_TIMERS)[threadIdx.x];
    }
    if constexpr (DBG & 128) {   // timing experiment: (input-staging cycles, total cycles, passes) of this wave into net_out
        if (lane == 0) {
            float *o = p.net_out + (size_t)(blockIdx.x * 4 + wave) * OUTC;
            o[0] = (float)t_stage; o[1] = (float)(__builtin_readcyclecounter() - t_kernel0); o[2] = (float)n_pass;
        }
    }
}

// =====================================================================================================
// Sky MLP: SKYMLP.forward on PE(raydir) (imaginaire/generators/gancraft_base.py:150-169; positional encoding
// .../voxlib/positional_encoding_kernel.cu:40-75) for every ray of the frame + the frame mean (scenedreamer.py:592-598)
// =====================================================================================================
// Same machinery as mlp_kernel (transposed register-resident chain, 3-term f16 split, LDS weight ring):
// 32 rays per wave, layers 33(->64 padded) -> 256 -> 256 x4 -> 64.  The style term fc_z_a(z) is folded into fc1's
// bias on the host.  The per-feature sum over rays (for sky_avg) is reduced per wave and written as one row of
// partial sums per wave (added up by the caller in a fixed order: reproducible, unlike float atomics).
constexpr int SKY_IN = 33, SKY_K0 = 64;                       // encoded ray direction, padded to 4 k-steps
constexpr int SKY_SLOTS = (4 + 4 * 16 + 4) * 4 / UNITS_PER_SLOT;   // 36
constexpr size_t SKY_L0_FRAGS = 16 * 4 * 64;                  // 16 units
constexpr size_t SKY_PACKED_FRAGS = SKY_L0_FRAGS + 4 * LH_FRAGS + LO_FRAGS;
constexpr int SC_BIAS1 = 0;                                   // [256] fc1.bias + fc_z_a(z)
constexpr int SC_BIASH = 256;                                 // [4][256] fc2..fc5 bias
constexpr int SC_BC = SC_BIASH + 4 * 256;                     // [64]
constexpr int SC_TOTAL = SC_BC + 64;

struct SkyParams {
    const float *raydirs;   // [R,3] ray directions, or (PRE) [R,33] rows that are already positional-encoded
    const half8 *wpk;
    const float *consts;    // SC_TOTAL floats
    float *sky_c;           // [R,64]
    float *sky_partial;     // [4 * gridDim.x][64]: every wave's sum of sky_c over its rays (summed by the caller: no float
                            // atomics, so the frame mean is reproducible bit for bit)
    float *sky_avg;         // optional [64]: frame mean of sky_c, finished by the last workgroup to arrive
    unsigned int *counter;  // with sky_avg: arrival counter, zero before the first launch (the kernel leaves it at zero)
    int32_t R, n_tiles;
};

struct SkyPackParams {
    const float *w1;        // [256,33]
    const float *wh[4];     // [256,256]
    const float *wc;        // [64,256]
    half8 *out;
};

__global__ __launch_bounds__(256) void sky_pack_kernel(const SkyPackParams p) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n0 = 16 * 2 * 64, nh = 64 * 2 * 64, no = 16 * 2 * 64;
    if (g >= n0 + 4 * nh + no) return;
    int layer, nib, ns, K;
    const float *W;
    size_t base, r = g;
    if (r < n0) {
        layer = 0; nib = 8; ns = 4; K = SKY_IN; W = p.w1; base = 0;
    } else if (r < n0 + 4 * nh) {
        r -= n0; layer = 1 + (int)(r / nh); r %= nh; nib = 8; ns = 16; K = HID; W = p.wh[layer - 1];
        base = SKY_L0_FRAGS + (size_t)(layer - 1) * LH_FRAGS;
    } else {
        r -= n0 + 4 * nh; layer = 5; nib = 2; ns = 16; K = HID; W = p.wc; base = SKY_L0_FRAGS + 4 * LH_FRAGS;
    }
    const int lane = (int)(r % 64); r /= 64;
    const int sel = (int)(r % 2);
    const int u = (int)(r / 2);
    int s, ib0;
    unit_coords(nib, ns, u, s, ib0);
    const int row = 32 * (ib0 + sel) + (lane & 31), h = lane >> 5;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int k = layer == 0 ? kmap_first(s, h, e) : kmap_hidden(s, h, e);
        float v = 0.f;
        if (k < K) v = W[(size_t)row * K + k] * (layer == 0 ? 1.0f : ACT_SCALE);
        const _Float16 vh = (_Float16)v;
        hi[e] = vh;
        lo[e] = (_Float16)(v - (float)vh);
    }
    p.out[base + ((size_t)u * 4 + 2 * sel + 0) * 64 + lane] = hi;
    p.out[base + ((size_t)u * 4 + 2 * sel + 1) * 64 + lane] = lo;
}

// element k of the positional encoding of direction d: [sin_0(3) cos_0(3) ... sin_4(3) cos_4(3) d(3)], zero padding
__device__ __forceinline__ float sky_pe(int k, float d0, float d1, float d2) {
    if (k >= SKY_IN) return 0.f;
    const int c = k % 3;
    const float x = c == 0 ? d0 : (c == 1 ? d1 : d2);
    if (k >= 30) return x;
    const int i = k / 6;
    const float rad = x * 3.141592654f * exp2f((float)i);    // positional_encoding_kernel.cu:63
    return ((k % 6) < 3) ? sinf(rad) : cosf(rad);
}

// SMX: the four hidden layers fc2..fc5 as f16 Whi.Xhi + fp6 corrections (layer8x; nothing amplifies the sky features' error)
// PRE: the input rows are SKYMLP.forward's own argument x [R,33] (the caller ran voxlib.positional_encoding, gancraft_base.py:150-157)
template <int DBG, int SMX, bool PRE = false>
__global__ __launch_bounds__(256, 1) void sky_kernel(const SkyParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_TOTAL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    float *cst = reinterpret_cast<float *>(lds + LDS_CONST);
    for (int i = threadIdx.x; i < SC_TOTAL; i += 256) cst[i] = p.consts[i];
    __syncthreads();

    Ring r;
    r.slots_per_pass = SKY_SLOTS;
    r.wbytes = reinterpret_cast<const char *>(p.wpk);
    r.g = 0;
    r.wave = __builtin_amdgcn_readfirstlane(wave);
    r.lane = lane;
    r.voff = r.wave * (PIECES * 1024) + lane * 16;
    r.lds_lane = (unsigned)(size_t)(const lds_char *)(lds + LDS_RING) + lane * 16;
    r.src_delta = r.wave * (PIECES * 1024) - (int)(unsigned)(size_t)(const lds_char *)(lds + LDS_RING);
#pragma unroll
    for (int sl = 0; sl < DMA_AHEAD; sl++) ring_issue(lds, r, sl, sl);
    r.next_in_pass = DMA_AHEAD;

    float fsum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // lane (q = j&3, h) owns features 32*ib + 8*q + 4*h + e
    const int q = j & 3;
    const int n_groups = (p.n_tiles + 3) >> 2;
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int tile = grp * 4 + wave;
        const int ray = tile * 32 + j;
        const bool ray_ok = tile < p.n_tiles && ray < p.R;
        const int rr = ray_ok ? ray : p.R - 1;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        if constexpr (!PRE) {
            d0 = p.raydirs[(size_t)rr * 3]; d1 = p.raydirs[(size_t)rr * 3 + 1]; d2 = p.raydirs[(size_t)rr * 3 + 2];
        }
        half8 bh[16], bl[16];
        f32x16 acc[8];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int k = 16 * s + 8 * h + e;
                if constexpr (PRE) v[e] = k < SKY_IN ? p.raydirs[(size_t)rr * SKY_IN + k] : 0.f;
                else v[e] = sky_pe(k, d0, d1, d2);
            }
            split8(v, bh[s], bl[s]);
        }
        float part = 0.f;
        const float *nul = cst;
        // fc1 (+ style term): 4 k-steps; fragments 0..2 of its upper half are activated behind its own lower half, the
        // remaining five (3..7) right after, the lower half behind fc2's head
        layer8<DBG, 4, false, false, false>(lds, r, bh, bl, acc, cst + SC_BIAS1, cst + SC_BIAS1, nul, h, part);
        act_step<3, false>(acc, cst + SC_BIAS1, nul, h, bh, bl, part);   // fragments 0..2 were activated inside the layer
        act_step<4, false>(acc, cst + SC_BIAS1, nul, h, bh, bl, part);
        act_step<5, false>(acc, cst + SC_BIAS1, nul, h, bh, bl, part);
        act_step<6, false>(acc, cst + SC_BIAS1, nul, h, bh, bl, part);
        act_step<7, false>(acc, cst + SC_BIAS1, nul, h, bh, bl, part);
        if constexpr (SMX) {
            // fc1's upper half (fragments 0..7 = K blocks 0, 1) was activated by the plain stages: block maxima from the
            // fragments, K block 0 converted here, K block 1 by fc2's first unit (the protocol of layer8x)
            MxState mx;
            mx.bm[2] = mx.bm[3] = 0.f;
            mx_block_max_f16<0>(bh, mx);
            mx_block_max_f16<1>(bh, mx);
            mx_convert<0>(bh, bl, mx);
#pragma unroll 1
            for (int l = 0; l < 4; l++) {
                const float *bias = cst + SC_BIASH + l * HID, *bias_pend = l == 0 ? cst + SC_BIAS1 : cst + SC_BIASH + (l - 1) * HID;
                if (l < 3) layer8x<DBG, 1, false, false>(lds, r, bh, bl, mx, acc, bias, bias_pend, nul, h, part);
                else layer8x<DBG, 2, false, false>(lds, r, bh, bl, mx, acc, bias, bias_pend, nul, h, part);
            }
        } else {
#pragma unroll 1
            for (int l = 0; l < 4; l++)
                layer8<DBG, 16, true, false, false>(lds, r, bh, bl, acc, cst + SC_BIASH + l * HID,
                                                    l == 0 ? cst + SC_BIAS1 : cst + SC_BIASH + (l - 1) * HID, nul, h, part);
        }
        f32x16 col[2];
        col[0] = bias_block<0>(cst + SC_BC, h);
        col[1] = bias_block<1>(cst + SC_BC, h);
        layer_out<DBG>(lds, r, bh, bl, acc, col, cst + SC_BIASH + 3 * HID, h, part);
        // ---- store sky_c[ray][feature] and accumulate the per-feature sum over rays -----------------------------
        if (ray_ok) {
#pragma unroll
            for (int ib = 0; ib < 2; ib++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++)
                    *reinterpret_cast<float4 *>(p.sky_c + (size_t)ray * OUTC + 32 * ib + 8 * g4 + 4 * h) =
                        make_float4(col[ib][4 * g4], col[ib][4 * g4 + 1], col[ib][4 * g4 + 2], col[ib][4 * g4 + 3]);
        }
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int rg = 0; rg < 16; rg++) {
                float v = ray_ok ? col[ib][rg] : 0.f;
                // sum over the 32 rays of this half-wave: within a row of 16 lanes by DPP (quad, half-row mirror, row mirror: every
                // lane ends up with the row's sum), ONE LDS exchange for the other row (five ds_bpermute per value before)
                v += quad_dpp<QUAD_XOR1>(v);
                v += quad_dpp<QUAD_XOR2>(v);
                v += quad_dpp<DPP_ROW_HALF_MIRROR>(v);
                v += quad_dpp<DPP_ROW_MIRROR>(v);
                v += __shfl_xor(v, 16);
                if ((rg >> 2) == q) fsum[ib][rg & 3] += v;
            }
    }
    // lanes with j < 4 (q = j) of each half hold the sums of features 32*ib + 8*q + 4*h + e
    if (j < 4) {
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                p.sky_partial[(size_t)(blockIdx.x * 4 + wave) * OUTC + 32 * ib + 8 * q + 4 * h + e] = fsum[ib][e];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- frame mean (scenedreamer.py:592-598): the last workgroup to arrive adds the partial rows of ALL workgroups in
    //      row order (fixed order, double accumulation: reproducible bit for bit, unlike float atomics) ----------------
    if (p.sky_avg == nullptr) return;
    int *ticket = reinterpret_cast<int *>(lds + LDS_FLAGS);
    double *red = reinterpret_cast<double *>(lds + LDS_RING);      // the weight ring is idle now
    __threadfence();                                               // this workgroup's rows are visible device-wide ...
    if (threadIdx.x == 0) *ticket = (int)atomicAdd(p.counter, 1u); // ... before its arrival is counted
    __syncthreads();
    if (*ticket != (int)gridDim.x - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int rows = 4 * (int)gridDim.x, f = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int r0 = part * (rows / 4), r1 = r0 + rows / 4;          // four consecutive quarters of the rows
    double acc_d = 0.0;
    for (int rw = r0; rw < r1; rw++) acc_d += (double)__builtin_nontemporal_load(p.sky_partial + (size_t)rw * OUTC + f);
    red[threadIdx.x] = acc_d;
    __syncthreads();
    if (threadIdx.x < OUTC) {
        const double tot = ((red[f] + red[64 + f]) + red[128 + f]) + red[192 + f];
        p.sky_avg[f] = (float)(tot / (double)p.R);
    }
    if (threadIdx.x == 0) *p.counter = 0u;                         // ready for the next launch
}

// =====================================================================================================
// debug probe: checks the MFMA operand layouts this file relies on (tests/test_fused_gpu.py)
// =====================================================================================================
__global__ void mfma_probe_kernel(const float *A, const float *B, float *C) {
    // A [32][16], B [16][32] row-major fp32 -> C [32][32]
    const int lane = threadIdx.x, h = lane >> 5, j = lane & 31;
    half8 a, b;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        a[e] = (_Float16)A[j * 16 + 8 * h + e];
        b[e] = (_Float16)B[(8 * h + e) * 32 + j];
    }
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = mfma16(a, b, c);
#pragma unroll
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = c[r];
}


// =====================================================================================================
// Render CNN tail as ONE register-resident chain:  conv4a -> LeakyReLU -> conv4b + y -> LeakyReLU -> conv4 -> tanh
// (RenderCNN.forward, imaginaire/generators/gancraft_base.py:219-225; tanh :603)
// =====================================================================================================
// The three 1x1 convolutions are a per-pixel MLP 256 -> 256 -> 256 -> 3, i.e. exactly what the layer machinery above evaluates
// for the field samples: 32 pixels per wave as MFMA columns, all 256 channels of a pixel in the wave's registers, weights
// through the LDS ring, 3-term f16 split.  As three conv_kernel launches (cnn.hip) the tail is bound by memory: it writes and
// re-reads the 256-channel activation twice (conv4a 0.30 ms + conv4b 0.42 ms per 960x540 frame for 2.2 GB); as a chain it reads
// the activation planes once and writes 3 floats per pixel.
//   input:   the running activation y as f16 hi / lo planes [16 chunks][Hb*Wb pixels][16 channels] (cnn.hip's layout).  A lane
//            loads its pixel's channels in the accumulator (C/D) order -- fragment T, element e = channel 16 T + (e & 3) +
//            8 (e >> 2) + 4 h: two 8-byte pieces per chunk and plane -- so conv4a's weights are packed like a hidden layer's
//            (kmap_hidden) and the RESIDUAL of conv4b is lane-local: the value added to accumulator register 8 Q + 4 HS + e of
//            row block IB is element 4 HS + e of the lane's own input fragment T = 2 IB + Q.  The input fragments are
//            overwritten by conv4a's activations, so a copy (yh / yl) stays live until conv4b's activation has consumed it;
//            hipcc parks what does not fit into the 256 VGPRs in the AGPRs the accumulators leave free.
//   layers:  conv4a = layer8 (upper half activated behind its own lower half, lower half behind conv4b's head);
//            conv4b = the same with act_stage_res for its own halves (bias + residual, then the shared stages);
//            conv4  = layer_out's units with conv4b's lower half as the pending work; rows 0..2 of row block 0 are the image.
constexpr int CHAIN_SLOTS = (64 + 64 + 16) / UNITS_PER_SLOT;    // 18 ring slots per 128 pixels
constexpr size_t CHAIN_FRAGS = 2 * LH_FRAGS + LO_FRAGS;
constexpr int CC_B4A = 0, CC_B4B = HID, CC_B4 = 2 * HID, CC_TOTAL = 2 * HID + OUTC;

struct ChainParams {
    const _Float16 *yh, *yl;   // input planes
    const half8 *wpk;          // conv4a | conv4b | conv4 (64 rows, 3 used) in the packed unit order
    const float *consts;       // CC_TOTAL floats: conv4a.bias | conv4b.bias | conv4.bias padded to 64
    float *img;                // [3][H*W]
    float *raw;                // optional [3][H*W]: conv4's output before tanh (RenderCNN.forward's return value, gancraft_base.py:221-225)
    int32_t H, W, Wb;
    long chunk_bytes;          // Hb*Wb*32: byte stride between channel chunks of a plane
    int32_t tiles_per_row, n_tiles;   // 32-pixel runs of one image row
};

struct ChainPackParams {
    const float *w4a, *w4b, *w4;   // [256,256], [256,256], [3,256]
    half8 *out;
};

__global__ __launch_bounds__(256) void chain_pack_kernel(const ChainPackParams p) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;   // one thread per (layer, unit, row block of the pair, lane)
    const size_t nh = 64 * 2 * 64, no = 16 * 2 * 64;
    if (g >= 2 * nh + no) return;
    const int layer = g < nh ? 0 : g < 2 * nh ? 1 : 2;
    size_t r = g - (size_t)layer * nh;
    const float *W = layer == 0 ? p.w4a : layer == 1 ? p.w4b : p.w4;
    const size_t base = (size_t)layer * LH_FRAGS;
    const int lane = (int)(r % 64); r /= 64;
    const int sel = (int)(r % 2);
    const int u = (int)(r / 2);
    int s, ib0;
    unit_coords(layer == 2 ? 2 : 8, 16, u, s, ib0);
    const int row = 32 * (ib0 + sel) + (lane & 31), h = lane >> 5;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        // conv4b / conv4 consume a' = LeakyReLU(x) / 0.4 (act_stage); conv4a consumes y itself
        const float w = (layer == 2 && row >= 3) ? 0.f : W[(size_t)row * HID + kmap_hidden(s, h, e)];
        const float v = w * (layer == 0 ? 1.0f : ACT_SCALE);
        const _Float16 vh = (_Float16)v;
        hi[e] = vh;
        lo[e] = (_Float16)(v - (float)vh);
    }
    p.out[base + ((size_t)u * 4 + 2 * sel + 0) * 64 + lane] = hi;
    p.out[base + ((size_t)u * 4 + 2 * sel + 1) * 64 + lane] = lo;
}

// act_stage with the residual: stage 1 adds the bias and y (hi + lo)
template <int T, int HS, int STAGE>
__device__ __forceinline__ void act_stage_res(const f32x16 (&acc)[8], const ActIn &in, half8 (&bh)[16], half8 (&bl)[16],
                                              const half8 (&yh)[16], const half8 (&yl)[16], float &part, ActRegs &g) {
    act_stage<T, HS, false, STAGE, true>(acc, in, bh, bl, part, g);
    if constexpr (STAGE == 1) {
#pragma unroll
        for (int e = 0; e < 4; e++) g.y[e] += (float)yh[T][4 * HS + e] + (float)yl[T][4 * HS + e];
    }
}

// conv4b: layer8_unit's 3-term path (half-rate ActPlan) with the residual in the activation of its OWN upper half
template <int DBG, int U>
__device__ __forceinline__ void chain_b_unit(char *lds, Ring &r, LayerState &st, half8 (&bh)[16], half8 (&bl)[16], f32x16 (&acc)[8],
                                             const half8 (&yh)[16], const half8 (&yl)[16], const float *bias,
                                             const float *bias_pend, int h, float &part) {
    constexpr int UNITS = 64, RD = RING_DEPTH, UPS = UNITS_PER_SLOT;
    using P = ActPlan<DBG, 16, true, false, false, U, true>;
    if constexpr (U % UPS == 0 && U != 0) {
        st.pos_cur = ring_acquire<DBG>(lds, r);
        st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    }
    constexpr int UN = U + RD - 1;
    constexpr bool PF = UN < UNITS;
    const int pf_pos = (UN / UPS) == (U / UPS) ? st.pos_cur : st.pos_nxt;
    constexpr int S = P::S, IB = 4 * P::HALF + 2 * (P::REM & 1);
    half8(&a)[4] = st.ring[U % RD];
    half8(&nx)[4] = st.ring[UN % RD];
    const ActIn &in = st.in[P::SLOT];
    constexpr bool PF_PREV = U == 0 || (U - 1 + RD - 1) < UNITS;
    lds_wait<PF_PREV ? 4 : 0>();
    layer8_fetch<DBG, 16, true, false, false, U + 1, true>(bias, bias_pend, bias, h, st);
#define SDN_STAGE(K) \
    if constexpr (U % UPS < PIECES / 4 && K < 4) ring_issue_piece<4 * (U % UPS) + ((K) & 3)>(lds, r); \
    if constexpr (P::stage(K) >= 0 && P::PEND) act_stage<P::T, P::HS, false, P::stage(K) < 0 ? 0 : P::stage(K), true>(acc, in, bh, bl, part, st.g); \
    if constexpr (P::stage(K) >= 0 && P::OWN) act_stage_res<P::T, P::HS, P::stage(K) < 0 ? 0 : P::stage(K)>(acc, in, bh, bl, yh, yl, part, st.g); \
    if constexpr (PF && K < 4) lds_frag<UN % UPS, (K) & 3>(r, pf_pos, nx[(K) & 3]); \
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) acc[IB] = mfma16(a[0], bh[S], zero16());
    else acc[IB] = mfma16(a[0], bh[S], acc[IB]);
    SDN_STAGE(0)
    if constexpr (S == 0) acc[IB + 1] = mfma16(a[2], bh[S], zero16());
    else acc[IB + 1] = mfma16(a[2], bh[S], acc[IB + 1]);
    SDN_STAGE(1)
    acc[IB] = mfma16(a[1], bh[S], acc[IB]);
    SDN_STAGE(2)
    acc[IB + 1] = mfma16(a[3], bh[S], acc[IB + 1]);
    SDN_STAGE(3)
    acc[IB] = mfma16(a[0], bl[S], acc[IB]);
    SDN_STAGE(4)
    acc[IB + 1] = mfma16(a[2], bl[S], acc[IB + 1]);
    SDN_STAGE(5)
#undef SDN_STAGE
}

template <int DBG, int... Us>
__device__ __forceinline__ void chain_b_units(std::integer_sequence<int, Us...>, char *lds, Ring &r, LayerState &st, half8 (&bh)[16],
                                              half8 (&bl)[16], f32x16 (&acc)[8], const half8 (&yh)[16], const half8 (&yl)[16],
                                              const float *bias, const float *bias_pend, int h, float &part) {
    (chain_b_unit<DBG, Us>(lds, r, st, bh, bl, acc, yh, yl, bias, bias_pend, h, part), ...);
}

template <int DBG>
__device__ __forceinline__ void chain_layer_b(char *lds, Ring &r, half8 (&bh)[16], half8 (&bl)[16], f32x16 (&acc)[8],
                                              const half8 (&yh)[16], const half8 (&yl)[16], const float *bias, const float *bias_pend,
                                              int h, float &part) {
    LayerState st;
    st.pos_cur = ring_acquire<DBG>(lds, r);
    st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    layer8_fetch<DBG, 16, true, false, false, 0, true>(bias, bias_pend, bias, h, st);
    lds_unit<0>(r, st.pos_cur, st.ring[0]);
    lds_unit<1>(r, st.pos_cur, st.ring[1]);
    chain_b_units<DBG>(std::make_integer_sequence<int, 64>{}, lds, r, st, bh, bl, acc, yh, yl, bias, bias_pend, h, part);
}

// a lane's input fragment T of one plane: channels 16 T + 4 h + {0..3} and 16 T + 8 + 4 h + {0..3} of its pixel (two 8-byte pieces)
struct ChainSrc {
    const char *h, *l;   // hi / lo plane + this lane's byte offset inside chunk 0
    long chunk_bytes;
};
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
template <int T>
__device__ __forceinline__ void chain_load(const ChainSrc &src, half8 &fh, half8 &fl) {
    const char *ph = src.h + (long)T * src.chunk_bytes, *pl = src.l + (long)T * src.chunk_bytes;
    const u32x2v h0 = *reinterpret_cast<const u32x2v *>(ph), h1 = *reinterpret_cast<const u32x2v *>(ph + 16);
    const u32x2v l0 = *reinterpret_cast<const u32x2v *>(pl), l1 = *reinterpret_cast<const u32x2v *>(pl + 16);
    fh = __builtin_bit_cast(half8, u32x4v{h0[0], h0[1], h1[0], h1[1]});
    fl = __builtin_bit_cast(half8, u32x4v{l0[0], l0[1], l1[0], l1[1]});
}

template <int... Ts>
__device__ __forceinline__ void chain_load_all(std::integer_sequence<int, Ts...>, const ChainSrc &src, half8 (&fh)[16], half8 (&fl)[16]) {
    (chain_load<Ts>(src, fh[Ts], fl[Ts]), ...);
}

// conv4: out_unit with the residual in the pending activation (conv4b's lower half).  The NEXT 128 pixels' input is loaded
// here, a layer ahead of its first use: unit U consumes fragment U for the last time, so fragment U of the next pass goes
// out at unit U + 1 (U < 8), and fragments 8..15 go out at units 0..7 into the registers the residual copy of fragments 0..7
// left free after conv4b.  All 64 loads are in flight by unit 8; hipcc waits for them (vmcnt(0): it cannot count across the
// loop's back edge) in front of conv4a's first MFMA, ~8 units later.
template <int DBG, int U>
__device__ __forceinline__ void chain_out_unit(char *lds, Ring &r, OutState &st, half8 (&bh)[16], half8 (&bl)[16], const f32x16 (&acc)[8],
                                               const half8 (&yh)[16], const half8 (&yl)[16], f32x16 (&col)[2], const float *bias_pend,
                                               int h, float &part, const ChainSrc &nsrc, half8 (&nh)[16], half8 (&nl)[16]) {
    constexpr int UNITS = 16, RD = RING_DEPTH, UPS = UNITS_PER_SLOT;
    if constexpr (U % UPS == 0 && U != 0) {
        st.pos_cur = ring_acquire<DBG>(lds, r);
        st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    }
    constexpr int UN = U + RD - 1;
    constexpr bool PF = UN < UNITS;
    const int pf_pos = (UN / UPS) == (U / UPS) ? st.pos_cur : st.pos_nxt;
    using P = OutPlan<DBG, U>;
    constexpr bool ACT = P::ACT, TWO = P::TWO;
    constexpr int T = P::T, HS = P::HS;
    ActRegs g0, g1;
    half8(&a)[4] = st.ring[U % RD];
    half8(&nx)[4] = st.ring[UN % RD];
    const ActIn &in0 = st.in[U & 1][0], &in1 = st.in[U & 1][1];
    constexpr bool PF_PREV = U == 0 || (U - 1 + RD - 1) < UNITS;
    lds_wait<PF_PREV ? 4 : 0>();
    out_fetch<DBG, U + 1>(bias_pend, h, st);
#define SDN_STAGE(K) \
    if constexpr (U % UPS < PIECES / 4 && K < 4) ring_issue_piece<4 * (U % UPS) + ((K) & 3)>(lds, r); \
    if constexpr (ACT) act_stage_res<T, HS, K>(acc, in0, bh, bl, yh, yl, part, g0); \
    if constexpr (TWO) act_stage_res<T, 1, K>(acc, in1, bh, bl, yh, yl, part, g1); \
    if constexpr (PF && K < 4) lds_frag<UN % UPS, (K) & 3>(r, pf_pos, nx[(K) & 3]); \
    if constexpr (U >= 1 && U <= 8 && K == 5) chain_load<(U >= 1 && U <= 8 ? U - 1 : 0)>(nsrc, nh[U >= 1 && U <= 8 ? U - 1 : 0], nl[U >= 1 && U <= 8 ? U - 1 : 0]); \
    if constexpr (U <= 7 && K == 4) chain_load<(U <= 7 ? 8 + U : 8)>(nsrc, nh[U <= 7 ? 8 + U : 8], nl[U <= 7 ? 8 + U : 8]); \
    __builtin_amdgcn_sched_barrier(0);
    col[0] = mfma16(a[0], bh[U], col[0]);
    SDN_STAGE(0)
    col[1] = mfma16(a[2], bh[U], col[1]);
    SDN_STAGE(1)
    col[0] = mfma16(a[1], bh[U], col[0]);
    SDN_STAGE(2)
    col[1] = mfma16(a[3], bh[U], col[1]);
    SDN_STAGE(3)
    col[0] = mfma16(a[0], bl[U], col[0]);
    SDN_STAGE(4)
    col[1] = mfma16(a[2], bl[U], col[1]);
    SDN_STAGE(5)
#undef SDN_STAGE
}

template <int DBG, int... Us>
__device__ __forceinline__ void chain_out_units(std::integer_sequence<int, Us...>, char *lds, Ring &r, OutState &st, half8 (&bh)[16],
                                                half8 (&bl)[16], const f32x16 (&acc)[8], const half8 (&yh)[16], const half8 (&yl)[16],
                                                f32x16 (&col)[2], const float *bias_pend, int h, float &part, const ChainSrc &nsrc,
                                                half8 (&nh)[16], half8 (&nl)[16]) {
    (chain_out_unit<DBG, Us>(lds, r, st, bh, bl, acc, yh, yl, col, bias_pend, h, part, nsrc, nh, nl), ...);
}

template <int DBG>
__device__ __forceinline__ void chain_layer_out(char *lds, Ring &r, half8 (&bh)[16], half8 (&bl)[16], const f32x16 (&acc)[8],
                                                const half8 (&yh)[16], const half8 (&yl)[16], f32x16 (&col)[2], const float *bias_pend,
                                                int h, float &part, const ChainSrc &nsrc, half8 (&nh)[16], half8 (&nl)[16]) {
    OutState st;
    st.pos_cur = ring_acquire<DBG>(lds, r);
    st.pos_nxt = (st.pos_cur + 1) & (NSLOT - 1);
    out_fetch<DBG, 0>(bias_pend, h, st);
    lds_unit<0>(r, st.pos_cur, st.ring[0]);
    lds_unit<1>(r, st.pos_cur, st.ring[1]);
    chain_out_units<DBG>(std::make_integer_sequence<int, 16>{}, lds, r, st, bh, bl, acc, yh, yl, col, bias_pend, h, part, nsrc, nh, nl);
}

__global__ __launch_bounds__(256, 1) void chain_kernel(const ChainParams p) {
    constexpr int DBG = 0;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_TOTAL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    float *cst = reinterpret_cast<float *>(lds + LDS_CONST);
    for (int i = threadIdx.x; i < CC_TOTAL; i += 256) cst[i] = p.consts[i];
    __syncthreads();

    Ring r;
    r.slots_per_pass = CHAIN_SLOTS;
    r.wbytes = reinterpret_cast<const char *>(p.wpk);
    r.g = 0;
    r.wave = __builtin_amdgcn_readfirstlane(wave);
    r.lane = lane;
    r.voff = r.wave * (PIECES * 1024) + lane * 16;
    r.lds_lane = (unsigned)(size_t)(const lds_char *)(lds + LDS_RING) + lane * 16;
    r.src_delta = r.wave * (PIECES * 1024) - (int)(unsigned)(size_t)(const lds_char *)(lds + LDS_RING);
#pragma unroll
    for (int sl = 0; sl < DMA_AHEAD; sl++) ring_issue(lds, r, sl, sl);
    r.next_in_pass = DMA_AHEAD;

    const int n_groups = (p.n_tiles + 3) >> 2;
    // where a group's pixels are: 32 consecutive x of one image row per wave; lanes beyond the row's end (and waves beyond the
    // last tile) evaluate a clamped pixel and store nothing
    struct Where { int y, x; bool ok; ChainSrc src; };
    auto where = [&](int grp) {
        Where w;
        const int tile = grp * 4 + wave;
        const bool tile_ok = tile < p.n_tiles;
        const int t = tile_ok ? tile : p.n_tiles - 1;
        w.y = t / p.tiles_per_row;
        w.x = (t - w.y * p.tiles_per_row) * 32 + j;
        w.ok = tile_ok && w.x < p.W;
        const int xc = w.x < p.W ? w.x : p.W - 1;
        // byte offset of this lane's first 8-byte piece inside chunk 0 (pixel (y, x) of the frame is buffer pixel (y+1, x+1))
        const long off = ((long)(w.y + 1) * p.Wb + (xc + 1)) * 32 + 8 * h;
        w.src.h = reinterpret_cast<const char *>(p.yh) + off;
        w.src.l = reinterpret_cast<const char *>(p.yl) + off;
        w.src.chunk_bytes = p.chunk_bytes;
        return w;
    };
    half8 bh[16], bl[16];
    Where cur = where(blockIdx.x < n_groups ? blockIdx.x : 0);
    chain_load_all(std::make_integer_sequence<int, 16>{}, cur.src, bh, bl);
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int grp_n = grp + (int)gridDim.x;
        const Where nxt = where(grp_n < n_groups ? grp_n : grp);   // (the last pass re-loads its own pixels: no branch in the loads)
        half8 yh[16], yl[16], nh[16], nl[16];
        f32x16 acc[8];
#pragma unroll
        for (int T = 0; T < 16; T++) { yh[T] = bh[T]; yl[T] = bl[T]; }
        float part = 0.f;
        // conv4a: its upper half is activated behind its own lower half, its lower half behind conv4b's head
        layer8<DBG, 16, false, false, false>(lds, r, bh, bl, acc, cst + CC_B4A, cst + CC_B4A, cst, h, part);
        // conv4b (+ y): the same, every activation of ITS outputs with the residual
        chain_layer_b<DBG>(lds, r, bh, bl, acc, yh, yl, cst + CC_B4B, cst + CC_B4A, h, part);
        f32x16 col[2];
        col[0] = bias_block<0>(cst + CC_B4, h);
        col[1] = bias_block<1>(cst + CC_B4, h);
        chain_layer_out<DBG>(lds, r, bh, bl, acc, yh, yl, col, cst + CC_B4B, h, part, nxt.src, nh, nl);
        asm volatile("" ::"v"(col[1]));   // (row block 1 of the projection is padding)
        // rows 0..2 of row block 0 = registers 0..2 of the h = 0 half: the image, gancraft_base.py:603
        if (cur.ok && h == 0) {
            const size_t o = (size_t)cur.y * p.W + cur.x;
#pragma unroll
            for (int c = 0; c < 3; c++) p.img[(size_t)c * p.H * p.W + o] = tanhf(col[0][c]);
            if (p.raw) {
#pragma unroll
                for (int c = 0; c < 3; c++) p.raw[(size_t)c * p.H * p.W + o] = col[0][c];
            }
        }
        cur.y = nxt.y; cur.x = nxt.x; cur.ok = nxt.ok;
#pragma unroll
        for (int T = 0; T < 16; T++) { bh[T] = nh[T]; bl[T] = nl[T]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring runs DMA_AHEAD slots ahead: let it land before the LDS is released
    __builtin_amdgcn_s_barrier();
}


// =====================================================================================================
// Render CNN head: net_out rows -> conv1 (1x1, 64 -> 256) -> LeakyReLU -> y as f16 hi / lo planes, in ONE kernel
// (RenderCNN.forward, gancraft_base.py:206; replaces planes_kernel + conv_kernel<1> of cnn.hip)
// =====================================================================================================
// Bound by writing y (0.54 GB per 548 x 968 frame); as two launches the 64-channel input was also written and re-read as planes
// and the weights went through conv_kernel's k loop for 4 k-steps per patch.  Here a wave takes 32 pixels: its lanes read their
// pixel's 64 floats straight from the fp32 rows (kmap_first order: 32 contiguous bytes per k-step and lane half), one layer8
// of 4 k-steps WITHOUT activation stages evaluates all 256 outputs, and the epilogue adds the bias, applies LeakyReLU and
// stores.  The output rows are permuted in the packed weights so that register r of lane half h of row block IB is channel
// 32 IB + 16 h + r: a lane owns the whole 16-channel chunk 2 IB + h of its pixel = 32 contiguous bytes of each plane.
constexpr int HEAD_K = 64, HEAD_NS = HEAD_K / 16, HEAD_UNITS = HEAD_NS * 4, HEAD_SLOTS = HEAD_UNITS / UNITS_PER_SLOT;   // 16 units, 2 slots
constexpr size_t HEAD_FRAGS = (size_t)HEAD_UNITS * 4 * 64;

struct HeadParams {
    const float *x;            // [H*W][64] fp32 rows
    const half8 *wpk;
    const float *bias;         // [256]
    _Float16 *oh, *ol;         // output planes [16][Hb*Wb][16]
    int32_t H, W, Wb;
    long chunk_elems;          // Hb*Wb*16: element stride between channel chunks of a plane
    int32_t tiles_per_row, n_tiles;
};

struct HeadPackParams {
    const float *w1;           // [256, 64]
    half8 *out;
};

__global__ __launch_bounds__(256) void head_pack_kernel(const HeadPackParams p) {
    const int g = blockIdx.x * 256 + threadIdx.x;   // one thread per (unit, row block of the pair, lane)
    if (g >= HEAD_UNITS * 2 * 64) return;
    const int lane = g % 64, sel = (g / 64) % 2, u = g / 128;
    int s, ib0;
    unit_coords(8, HEAD_NS, u, s, ib0);
    // MFMA row rho of the block = register (rho & 3) + 4 (rho >> 3) of lane half (rho >> 2) & 1  ->  channel 32 IB + 16 h + r
    const int rho = lane & 31, ib = ib0 + sel;
    const int row = 32 * ib + 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
    const int h = lane >> 5;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float v = p.w1[(size_t)row * HEAD_K + kmap_first(s, h, e)];
        const _Float16 vh = (_Float16)v;
        hi[e] = vh;
        lo[e] = (_Float16)(v - (float)vh);
    }
    p.out[((size_t)u * 4 + 2 * sel + 0) * 64 + lane] = hi;
    p.out[((size_t)u * 4 + 2 * sel + 1) * 64 + lane] = lo;
}

__global__ __launch_bounds__(256, 1) void head_kernel(const HeadParams p) {
    constexpr int NOACT = 4;   // layer8's switch for "no activation stages" (the ablation bit): the epilogue below is the activation
    __shared__ __attribute__((aligned(1024))) char lds[LDS_TOTAL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    float *cst = reinterpret_cast<float *>(lds + LDS_CONST);
    for (int i = threadIdx.x; i < HID; i += 256) cst[i] = p.bias[i];
    __syncthreads();

    Ring r;
    r.slots_per_pass = HEAD_SLOTS;
    r.wbytes = reinterpret_cast<const char *>(p.wpk);
    r.g = 0;
    r.wave = __builtin_amdgcn_readfirstlane(wave);
    r.lane = lane;
    r.voff = r.wave * (PIECES * 1024) + lane * 16;
    r.lds_lane = (unsigned)(size_t)(const lds_char *)(lds + LDS_RING) + lane * 16;
    r.src_delta = r.wave * (PIECES * 1024) - (int)(unsigned)(size_t)(const lds_char *)(lds + LDS_RING);
#pragma unroll
    for (int sl = 0; sl < DMA_AHEAD; sl++) ring_issue(lds, r, sl, sl % HEAD_SLOTS);   // (the stream of a pass is 2 slots: it wraps)
    r.next_in_pass = DMA_AHEAD % HEAD_SLOTS;

    const int n_groups = (p.n_tiles + 3) >> 2;
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int tile = grp * 4 + wave;
        const bool tile_ok = tile < p.n_tiles;
        const int t = tile_ok ? tile : p.n_tiles - 1;
        const int y = t / p.tiles_per_row, x = (t - y * p.tiles_per_row) * 32 + j;
        const bool ok = tile_ok && x < p.W;
        const int xc = x < p.W ? x : p.W - 1;
        const float *src = p.x + ((size_t)y * p.W + xc) * HEAD_K + 8 * h;
        half8 bh[16], bl[16];
        f32x16 acc[8];
#pragma unroll
        for (int s = 0; s < HEAD_NS; s++) {
            const float4 a = *reinterpret_cast<const float4 *>(src + 16 * s), b = *reinterpret_cast<const float4 *>(src + 16 * s + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            split8(v, bh[s], bl[s]);
        }
        float part = 0.f;
        layer8<NOACT, HEAD_NS, false, false, false>(lds, r, bh, bl, acc, cst, cst, cst, h, part);
        // ---- bias, LeakyReLU, f16 hi / lo split, stores: register r of row block ib is channel 32 ib + 16 h + r of this lane's pixel
        const long pix = ((long)(y + 1) * p.Wb + (xc + 1)) * 16;
#pragma unroll
        for (int ib = 0; ib < 8; ib++) {
            const float *bsrc = cst + 32 * ib + 16 * h;
            half8 hv[2], lv[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float4 b0 = *reinterpret_cast<const float4 *>(bsrc + 8 * q), b1 = *reinterpret_cast<const float4 *>(bsrc + 8 * q + 4);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float t0 = acc[ib][8 * q + e] + bb[e];
                    v[e] = vmax_raw(t0, 0.2f * t0);   // LeakyReLU(0.2), as conv_kernel's epilogue
                }
                split8(v, hv[q], lv[q]);
            }
            if (ok) {
                _Float16 *oh = p.oh + (long)(2 * ib + h) * p.chunk_elems + pix, *ol = p.ol + (long)(2 * ib + h) * p.chunk_elems + pix;
                *reinterpret_cast<half8 *>(oh) = hv[0];
                *reinterpret_cast<half8 *>(oh + 8) = hv[1];
                *reinterpret_cast<half8 *>(ol) = lv[0];
                *reinterpret_cast<half8 *>(ol + 8) = lv[1];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring runs DMA_AHEAD slots ahead: let it land before the LDS is released
    __builtin_amdgcn_s_barrier();
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

size_t sdn_field_packed_weight_bytes(void) { return PACKED_FRAGS * sizeof(half8); }
size_t sdn_field_consts_floats(void) { return C_TOTAL; }
int sdn_field_const_offset(int which) {
    switch (which) {
        case 0: return C_LABEL_BIAS;
        case 1: return C_BETA;
        case 2: return C_WSIGMA;
        case 3: return C_BC;
        case 4: return C_BSIGMA;
        case 5: return C_SKY_AVG;
        default: return -1;
    }
}

int sdn_field_collapse_table(const float *embeddings, const int32_t *offsets_host, uint32_t L, float S, uint32_t H,
                             const float *genc_host, float *table3, sdn_stream_t stream) {
    SDN_REQUIRE(embeddings && offsets_host && genc_host && table3, "sdn_field_collapse_table: null pointer");
    if (L != NLEV) return sdn::fail(SDN_ERR_UNSUPPORTED, "fused field path needs the 16-level SceneDreamer grid");
    CollapseParams p;
    p.emb = embeddings;
    p.table3 = table3;
    const uint32_t T = (uint32_t)(offsets_host[1] - offsets_host[0]);
    if (T == 0 || (T & (T - 1)) != 0)
        return sdn::fail(SDN_ERR_UNSUPPORTED, "fused field path needs a power-of-two hash table per level");
    p.T = T;
    for (uint32_t l = 0; l < L; l++) {
        if ((uint32_t)(offsets_host[l + 1] - offsets_host[l]) != T)
            return sdn::fail(SDN_ERR_UNSUPPORTED, "fused field path needs equally sized levels");
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1;
        // every level must take the hash branch of get_grid_index (gridencoder.cu:60-69) on the 5-D grid
        double stride = 1;
        for (int d = 0; d < 5 && stride <= (double)T; d++) stride *= (double)(res + 1);
        if (!(stride > (double)T))
            return sdn::fail(SDN_ERR_UNSUPPORTED, "fused field path needs every level hashed (level %u is dense)", l);
        p.off[l] = (uint32_t)offsets_host[l];
        float fr[2];
        uint32_t pg[2];
        for (int d = 0; d < 2; d++) {
            const float x = (genc_host[d] + 1.f) / 2.f;  // grid.py:144
            volatile float prod = x * scale;             // two roundings, like the kernels
            const float pos = prod + 0.5f;
            const float fl = floorf(pos);
            pg[d] = (uint32_t)fl;
            fr[d] = pos - fl;
        }
        for (int c = 0; c < 4; c++) {
            const int c3 = c & 1, c4 = c >> 1;
            const uint32_t k = ((pg[0] + c3) * 3674653429u) ^ ((pg[1] + c4) * 2097192037u);
            p.K[l][c] = k & (T - 1);
            // reference multiply order: ((((1*w0)*w1)*w2)*w3)*w4 -> here the trailing w3*w4 factor
            volatile float w3 = c3 ? fr[0] : 1.f - fr[0];
            volatile float w4 = c4 ? fr[1] : 1.f - fr[1];
            p.w[l][c] = w3 * w4;
        }
    }
    hipLaunchKernelGGL(collapse_kernel, dim3(sdn::div_up<uint32_t>(T, 256), L), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_field_collapse_table");
}

int sdn_field_trunk_shift(void) { return TRUNK_SHIFT; }

int sdn_field_pack_weights(const float *w1, const float *const *wh5_host, const float *wc, void *packed,
                           sdn_stream_t stream) {
    SDN_REQUIRE(w1 && wh5_host && wc && packed, "sdn_field_pack_weights: null pointer");
    PackParams p;
    p.w1 = w1;
    for (int i = 0; i < 5; i++) {
        SDN_REQUIRE(wh5_host[i], "sdn_field_pack_weights: null hidden weight");
        p.wh[i] = wh5_host[i];
    }
    p.wc = wc;
    p.out = (half8 *)packed;
    const size_t n = 8 * 8 * 64 + 5 * 16 * 8 * 64 + 16 * 2 * 64;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)sdn::div_up<size_t>(n, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_field_pack_weights");
}

int sdn_field_pack_weights_mx(const float *w1, const float *const *wh5_host, const float *wc, void *packed, sdn_stream_t stream) {
    if (int rc = sdn_field_pack_weights(w1, wh5_host, wc, packed, stream)) return rc;
    PackMxParams p;
    p.wh[0] = wh5_host[3];   // fc_5
    p.wh[1] = wh5_host[4];   // fc_6
    p.wh[2] = p.wh[3] = nullptr;
    p.n_layers = 2;
    p.base = L0_FRAGS + 3 * LH_FRAGS;
    p.out = (half8 *)packed;
    hipLaunchKernelGGL(pack_mx_kernel, dim3(2 * 64 * 64 / 256), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_field_pack_weights_mx");
}

size_t sdn_field_feat_bytes(int32_t n_rays, int32_t num_samples) {
    const size_t tiles = (size_t)sdn::div_up(n_rays, RAYS_PER_TILE), nch = (size_t)sdn::div_up(num_samples, SAMP_PER_STEP);
    return tiles * nch * 8 * 64 * 8 * sizeof(float);
}
size_t sdn_field_aux_elems(int32_t n_rays, int32_t num_samples) {
    const size_t tiles = (size_t)sdn::div_up(n_rays, RAYS_PER_TILE), nch = (size_t)sdn::div_up(num_samples, SAMP_PER_STEP);
    return tiles * nch * 32;
}

// window_host: NULL, or {n_src, pitch, first, cols, ray0} (see RayWindow)
// launch_rays (sdn_field_render only): receives the rays of the launch -- n_rays, or, for a ragged blocked window
// (window_host[5] == 2), 32 x the blocks of its ceil(cols / 8) x ceil(rows / 4) grid
static int set_window(RayWindow &w, const int32_t *window_host, int32_t n_rays, const char *who, int32_t *launch_rays = nullptr) {
    w.tiled_bx = 0;
    w.rows = 0;
    if (launch_rays) *launch_rays = n_rays;
    if (window_host == nullptr) {
        w.n_src = n_rays; w.pitch = 0; w.first = 0; w.cols = 0; w.ray0 = 0;
        return 0;
    }
    w.n_src = window_host[0]; w.pitch = window_host[1]; w.first = window_host[2]; w.cols = window_host[3]; w.ray0 = window_host[4];
    if (w.n_src <= 0 || w.cols < 0 || w.ray0 < 0 || w.first < 0 || w.pitch < 0)
        return sdn::fail(SDN_ERR_INVALID, "%s: bad ray window", who);
    if (window_host[5] == 2) {   // 8 x 4 pixel blocks over a whole window of any size: positions outside it are no rays
        if (!launch_rays) return sdn::fail(SDN_ERR_UNSUPPORTED, "%s: the ragged blocked ray order exists for sdn_field_render only", who);
        if (!(w.cols > 0 && w.ray0 == 0 && n_rays % w.cols == 0))
            return sdn::fail(SDN_ERR_INVALID, "%s: the blocked ray order needs a whole window (ray0 = 0, n_rays = rows x cols)", who);
        const int rows = n_rays / w.cols;
        w.tiled_bx = (w.cols + 7) / 8;
        if (w.cols % 8 || rows % 4) {
            w.rows = rows;
            const long lr = (long)w.tiled_bx * ((rows + 3) / 4) * 32;
            if (lr >= ((long)1 << 31)) return sdn::fail(SDN_ERR_INVALID, "%s: window too large", who);
            *launch_rays = (int32_t)lr;
        }
    } else if (window_host[5]) {   // 8 x 4 pixel blocks: the launch is the whole window, whole blocks only
        if (!(w.cols > 0 && w.cols % 8 == 0 && w.ray0 == 0 && n_rays % w.cols == 0 && (n_rays / w.cols) % 4 == 0))
            return sdn::fail(SDN_ERR_INVALID, "%s: the blocked ray order needs a whole window of 8k columns x 4m rows (ray0 = 0)", who);
        w.tiled_bx = w.cols / 8;
    }
    const long last = (long)w.ray0 + n_rays - 1;
    const long src_last = w.cols > 0 ? (long)w.first + (last / w.cols) * w.pitch + (w.cols - 1) : last;
    if (src_last >= w.n_src) return sdn::fail(SDN_ERR_INVALID, "%s: ray window reaches outside the %d source rays", who, w.n_src);
    return 0;
}

// fills the encode-stage parameters (shared by sdn_field_encode and sdn_field_render); outputs / window are set by the caller
static int fill_enc(EncParams &p, const char *who, const int32_t *voxel_id, const float *depth2, const float *raydirs,
                    const uint8_t *lut1024, const float *table3, uint32_t table_rows, const float *scales_dev, const float *genc_host,
                    const float *cam_ori_host, const float *voxel_dims_host, const float *lin_dev, const float *u_dev, int32_t n_rays,
                    int32_t max_blocks, int32_t num_samples, float sample_depth, float dists_scale, int32_t strat_division) {
    if (!(voxel_id && depth2 && raydirs && lut1024 && table3 && scales_dev && genc_host && cam_ori_host && voxel_dims_host && lin_dev))
        return sdn::fail(SDN_ERR_INVALID, "%s: null pointer", who);
    if (!(strat_division == 0 || strat_division == 1))
        return sdn::fail(SDN_ERR_INVALID, "%s: strat_division must be 0 (x * (1/n)) or 1 (x / n)", who);
    if (!(n_rays > 0 && num_samples > 0)) return sdn::fail(SDN_ERR_INVALID, "%s: empty frame", who);
    if (max_blocks < 1 || max_blocks > MAXM) return sdn::fail(SDN_ERR_UNSUPPORTED, "%s: max_blocks must be 1..8", who);
    if (num_samples + 1 > MAX_LIN) return sdn::fail(SDN_ERR_UNSUPPORTED, "%s: at most 79 samples per ray", who);
    if (!(table_rows && (table_rows & (table_rows - 1)) == 0)) return sdn::fail(SDN_ERR_INVALID, "%s: table_rows must be a power of two", who);
    p.voxel_id = voxel_id; p.depth2 = depth2; p.raydirs = raydirs; p.lut = lut1024; p.table3 = table3;
    p.feat = nullptr; p.dist = nullptr; p.label = nullptr; p.rayflag = nullptr;
    p.R = n_rays; p.M = max_blocks; p.ns = num_samples;
    p.nch = sdn::div_up(num_samples, SAMP_PER_STEP);
    p.n_tiles = sdn::div_up(n_rays, RAYS_PER_TILE);
    p.tmask = table_rows - 1;
    p.genc_oob = 0;
    for (int d = 0; d < 2; d++) {
        const float x = (genc_host[d] + 1.f) / 2.f;
        if (x < 0.f || x > 1.f) p.genc_oob = 1;
    }
    for (int i = 0; i < 3; i++) { p.ori[i] = cam_ori_host[i]; p.delim[i] = voxel_dims_host[i]; }
    p.sample_depth = sample_depth; p.dists_scale = dists_scale;
    p.lin = lin_dev;
    p.u = u_dev;
    p.ieee_div = strat_division;
    p.scales = scales_dev;
    return 0;
}

int sdn_field_encode(const int32_t *voxel_id, const float *depth2, const float *raydirs, const uint8_t *lut1024,
                     const float *table3, uint32_t table_rows, const float *scales_dev, const float *genc_host,
                     const float *cam_ori_host, const float *voxel_dims_host, const float *lin_dev, const float *u_dev,
                     int32_t n_rays, int32_t max_blocks, int32_t num_samples, float sample_depth, float dists_scale, float *feat,
                     float *dist, uint8_t *label, uint8_t *rayflag, const int32_t *window_host, int32_t strat_division,
                     sdn_stream_t stream) {
    SDN_REQUIRE(feat && dist && label && rayflag, "sdn_field_encode: null pointer");
    EncParams p;
    if (int rc = fill_enc(p, "sdn_field_encode", voxel_id, depth2, raydirs, lut1024, table3, table_rows, scales_dev, genc_host, cam_ori_host,
                          voxel_dims_host, lin_dev, u_dev, n_rays, max_blocks, num_samples, sample_depth, dists_scale, strat_division))
        return rc;
    p.feat = feat; p.dist = dist; p.label = label; p.rayflag = rayflag;
    if (int rc = set_window(p.win, window_host, n_rays, "sdn_field_encode")) return rc;
    hipLaunchKernelGGL(encode_kernel, dim3(sdn::div_up(p.n_tiles, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_field_encode");
}

int sdn_sample_depth(const float *depth2, const float *lin_dev, const float *u_dev, int32_t n_rays, int32_t max_blocks,
                     int32_t n_points, float sample_depth, float *rand_depth, float *new_dists, int64_t *idx,
                     int32_t strat_division, sdn_stream_t stream) {
    SDN_REQUIRE(depth2 && lin_dev && rand_depth && new_dists && idx, "sdn_sample_depth: null pointer");
    SDN_REQUIRE(strat_division == 0 || strat_division == 1, "sdn_sample_depth: strat_division must be 0 (x * (1/n)) or 1 (x / n)");
    SDN_REQUIRE(n_rays > 0 && n_points >= 2, "sdn_sample_depth: need at least one ray and two stratified points");
    if (max_blocks < 1 || max_blocks > MAXM) return sdn::fail(SDN_ERR_UNSUPPORTED, "sdn_sample_depth: max_blocks must be 1..8");
    SampleParams p;
    p.depth2 = depth2; p.lin = lin_dev; p.u = u_dev; p.rand_depth = rand_depth; p.new_dists = new_dists; p.idx = idx;
    p.R = n_rays; p.M = max_blocks; p.n_points = n_points; p.sample_depth = sample_depth; p.ieee_div = strat_division;
    hipLaunchKernelGGL(sample_depth_kernel, dim3(sdn::div_up(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_sample_depth");
}

static int fill_mlp(MlpParams &p, const char *who, const void *packed, const float *consts, const float *sky_c, float *net_out,
                    int32_t n_rays, int32_t num_samples, int32_t colour_terms, float term_eps, uint8_t *passes, const int32_t *window_host,
                    const float *sky_avg, int32_t *ticket, int32_t *launch_rays = nullptr) {
    if (!(packed && consts && sky_c && net_out)) return sdn::fail(SDN_ERR_INVALID, "%s: null pointer", who);
    if (!(n_rays > 0 && num_samples > 0)) return sdn::fail(SDN_ERR_INVALID, "%s: empty frame", who);
    if (!(colour_terms == 2 || colour_terms == 3 || colour_terms == 6)) return sdn::fail(SDN_ERR_INVALID, "%s: colour_terms must be 2, 3 or 6", who);
    if (!(term_eps >= 0.f && term_eps < 1.f)) return sdn::fail(SDN_ERR_INVALID, "%s: term_eps must be in [0, 1)", who);
    p.term_depth = term_eps > 0.f ? -logf(term_eps) : 0.f;
    p.passes = passes;
    p.cam_ori_dev = nullptr; p.w_out = nullptr; p.depth_out = nullptr; p.sigma_out = nullptr;
    p.sig_out = nullptr; p.col_out = nullptr; p.skyb_out = nullptr; p.nosky_out = nullptr;
    p.colour_passes = nullptr; p.no_colour_skip = 0;
    p.feat = nullptr; p.dist = nullptr; p.label = nullptr; p.rayflag = nullptr;
    p.wpk = (const half8 *)packed;
    p.consts = consts; p.sky_c = sky_c; p.net_out = net_out;
    p.sky_avg = sky_avg; p.ticket = ticket;
    if (int rc = set_window(p.win, window_host, n_rays, who, launch_rays)) return rc;
    p.R = n_rays; p.ns = num_samples;
    p.nch = sdn::div_up(num_samples, SAMP_PER_STEP);
    p.n_tiles = sdn::div_up(n_rays, RAYS_PER_TILE);
    return 0;
}

static int mlp_workgroups(const MlpParams &p, int32_t n_workgroups) {
    int wg = n_workgroups > 0 ? n_workgroups : 256;
    const int groups = sdn::div_up(p.n_tiles, 4);
    return wg > groups ? groups : wg;
}

int sdn_field_mlp(const float *feat, const float *dist, const uint8_t *label, const uint8_t *rayflag, const void *packed,
                  const float *consts, const float *sky_c, float *net_out, int32_t n_rays, int32_t num_samples,
                  int32_t colour_terms, float term_eps, uint8_t *passes, int32_t n_workgroups, const int32_t *window_host,
                  const float *sky_avg, int32_t *ticket, sdn_stream_t stream) {
    SDN_REQUIRE(feat && dist && label && rayflag, "sdn_field_mlp: null pointer");
    MlpParams p;
    if (int rc = fill_mlp(p, "sdn_field_mlp", packed, consts, sky_c, net_out, n_rays, num_samples, colour_terms, term_eps, passes, window_host,
                          sky_avg, ticket))
        return rc;
    p.feat = feat; p.dist = dist; p.label = label; p.rayflag = rayflag;
    const int wg = mlp_workgroups(p, n_workgroups);
    static const int dbg = [] {
        const char *e = getenv("SDN_MLP_DBG");   // timing experiments only; results are wrong unless 0
        return e ? atoi(e) : 0;
    }();
    switch (dbg) {
#ifdef SDN_MLP_ABLATION
        case 1: hipLaunchKernelGGL((mlp_kernel<1, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;   // no ring DMA
        case 256: hipLaunchKernelGGL((mlp_kernel<256, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break; // no input prefetch
        case 384: hipLaunchKernelGGL((mlp_kernel<384, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;
        case 128: hipLaunchKernelGGL((mlp_kernel<128, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break; // input-staging timer
        case 2: hipLaunchKernelGGL((mlp_kernel<2, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;   // no ring barrier
        case 3: hipLaunchKernelGGL((mlp_kernel<3, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;
        case 4: hipLaunchKernelGGL((mlp_kernel<4, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;   // no activation VALU
        case 8: hipLaunchKernelGGL((mlp_kernel<8, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;   // no fragment ds_read
        case 16: hipLaunchKernelGGL((mlp_kernel<16, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break; // no MFMA
        case 28: hipLaunchKernelGGL((mlp_kernel<28, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;
        case 32: hipLaunchKernelGGL((mlp_kernel<32, 6>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;  // colour layers without Wlo.X
        case 64: hipLaunchKernelGGL((mlp_kernel<64, 6>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;  // ... without Whi.Xlo
        case 96: hipLaunchKernelGGL((mlp_kernel<96, 6>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;  // ... Whi.Xhi only
        case 512: hipLaunchKernelGGL((mlp_kernel<512, 6>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break; // per-segment timers
        case 515: hipLaunchKernelGGL((mlp_kernel<512, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;
        case 513: hipLaunchKernelGGL((mlp_kernel<513, 6>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break; // timers + one ablation
        case 514: hipLaunchKernelGGL((mlp_kernel<514, 6>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;
        case 516: hipLaunchKernelGGL((mlp_kernel<516, 6>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;
        case 520: hipLaunchKernelGGL((mlp_kernel<520, 6>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); break;
#endif
        default:
            if (colour_terms == 2) hipLaunchKernelGGL((mlp_kernel<0, 2>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
            else if (colour_terms == 6) hipLaunchKernelGGL((mlp_kernel<0, 6>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
            else hipLaunchKernelGGL((mlp_kernel<0, 3>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
            break;
    }
    return sdn::check_launch("sdn_field_mlp");
}

int sdn_field_render(const int32_t *voxel_id, const float *depth2, const float *raydirs, const uint8_t *lut1024, const float *table3,
                     uint32_t table_rows, const float *scales_dev, const float *genc_host, const float *cam_ori_host,
                     const float *voxel_dims_host, const float *lin_dev, const float *u_dev, int32_t n_rays, int32_t max_blocks,
                     int32_t num_samples, float sample_depth, float dists_scale, const void *packed, const float *consts,
                     const float *sky_c, const float *sky_avg, float *net_out, int32_t colour_terms, float term_eps, uint8_t *passes,
                     int32_t n_workgroups, const int32_t *window_host, int32_t strat_division, int32_t *ticket, const float *cam_ori_dev,
                     const sdn_field_aux *aux, sdn_stream_t stream) {
    MlpParams p;
    const bool want_aux = aux && (aux->weights || aux->depth || aux->sigma || aux->colour || aux->sky_blended || aux->nosky);
    SDN_REQUIRE(!(want_aux && term_eps > 0.f), "sdn_field_render: the per-sample outputs need term_eps = 0 (every pass must run)");
    static const float zero3[3] = {0.f, 0.f, 0.f};
    if (cam_ori_dev && !cam_ori_host) cam_ori_host = zero3;
    int32_t launch_rays = n_rays;
    if (int rc = fill_mlp(p, "sdn_field_render", packed, consts, sky_c, net_out, n_rays, num_samples, colour_terms, term_eps, passes,
                          window_host, sky_avg, ticket, &launch_rays))
        return rc;
    if (int rc = fill_enc(p.enc, "sdn_field_render", voxel_id, depth2, raydirs, lut1024, table3, table_rows, scales_dev, genc_host,
                          cam_ori_host, voxel_dims_host, lin_dev, u_dev, n_rays, max_blocks, num_samples, sample_depth, dists_scale,
                          strat_division))
        return rc;
    p.enc.win = p.win;
    if (launch_rays != n_rays) {   // ragged blocked window: the launch walks the whole block grid (the extra positions are no rays)
        SDN_REQUIRE(u_dev == nullptr, "sdn_field_render: stochastic sampling with the ragged blocked order is not supported");
        p.R = p.enc.R = launch_rays;
        p.n_tiles = p.enc.n_tiles = launch_rays / RAYS_PER_TILE;
    }
    p.cam_ori_dev = cam_ori_dev;
    if (want_aux) {
        p.w_out = aux->weights; p.depth_out = aux->depth; p.sig_out = aux->sigma; p.col_out = aux->colour;
        p.skyb_out = aux->sky_blended; p.nosky_out = aux->nosky;
    }
    if (aux) {   // (these two do not select the per-sample-output instantiation)
        p.colour_passes = aux->colour_passes;
        p.no_colour_skip = (aux->flags & SDN_FIELD_NO_COLOUR_SKIP) ? 1 : 0;
    }
    SDN_REQUIRE(colour_terms != 2, "sdn_field_render: colour_terms must be 3 or 6 (the 2-term profile exists for sdn_field_mlp only)");
    const int wg = mlp_workgroups(p, n_workgroups);
#ifdef SDN_MLP_ABLATION
    if (const char *e = getenv("SDN_MLP_DBG")) {   // timing experiments only
        if (atoi(e) == 512) { hipLaunchKernelGGL((mlp_kernel<512, 6, MODE_FUSED>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); return sdn::check_launch("sdn_field_render"); }
        if (atoi(e) == 515) { hipLaunchKernelGGL((mlp_kernel<512, 3, MODE_FUSED>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p); return sdn::check_launch("sdn_field_render"); }
    }
#endif
    if (want_aux) {
        if (colour_terms == 6) hipLaunchKernelGGL((mlp_kernel<0, 6, MODE_FUSED_AUX>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((mlp_kernel<0, 3, MODE_FUSED_AUX>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    } else if (colour_terms == 6) hipLaunchKernelGGL((mlp_kernel<0, 6, MODE_FUSED>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((mlp_kernel<0, 3, MODE_FUSED>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_field_render");
}

// LightningMLP.forward as an op (imaginaire/model_utils/layers.py:92-126, N = 1 so the ModLinear modulation is folded into
// the packed weights / consts exactly as for sdn_field_render)
int sdn_render_mlp(const float *x, const uint8_t *label, const void *packed, const float *consts, float *sigma, float *c,
                   int64_t n_rows, int32_t colour_terms, int32_t n_workgroups, int32_t *ticket, sdn_stream_t stream) {
    SDN_REQUIRE(x && label && packed && consts && sigma && c, "sdn_render_mlp: null pointer");
    SDN_REQUIRE(n_rows > 0 && n_rows < ((int64_t)1 << 31), "sdn_render_mlp: n_rows must be in [1, 2^31)");
    SDN_REQUIRE(colour_terms == 3 || colour_terms == 6, "sdn_render_mlp: colour_terms must be 3 or 6");
    MlpParams p;
    p.feat = x; p.dist = nullptr; p.label = label; p.rayflag = nullptr;
    p.wpk = (const half8 *)packed; p.consts = consts; p.sky_c = nullptr; p.net_out = c;
    p.R = (int32_t)n_rows; p.ns = 32; p.nch = 8;
    p.n_tiles = (int32_t)((n_rows + 255) / 256);
    p.term_depth = 0.f; p.passes = nullptr;
    p.win.n_src = p.R; p.win.pitch = 0; p.win.first = 0; p.win.cols = 0; p.win.ray0 = 0; p.win.tiled_bx = 0; p.win.rows = 0;
    p.sky_avg = nullptr; p.ticket = ticket;
    p.cam_ori_dev = nullptr; p.w_out = nullptr; p.depth_out = nullptr; p.sigma_out = sigma;
    p.sig_out = nullptr; p.col_out = nullptr; p.skyb_out = nullptr; p.nosky_out = nullptr;
    p.colour_passes = nullptr; p.no_colour_skip = 0;
    p.enc = EncParams{};
    const int wg = mlp_workgroups(p, n_workgroups);
    if (colour_terms == 6) hipLaunchKernelGGL((mlp_kernel<0, 6, MODE_RAW>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((mlp_kernel<0, 3, MODE_RAW>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_render_mlp");
}

size_t sdn_conv_chain_packed_weight_bytes(void) { return CHAIN_FRAGS * sizeof(half8); }
size_t sdn_conv_chain_consts_floats(void) { return CC_TOTAL; }

int sdn_conv_chain_pack_weights(const float *w4a, const float *w4b, const float *w4, void *packed, sdn_stream_t stream) {
    SDN_REQUIRE(w4a && w4b && w4 && packed, "sdn_conv_chain_pack_weights: null pointer");
    ChainPackParams p;
    p.w4a = w4a; p.w4b = w4b; p.w4 = w4; p.out = (half8 *)packed;
    const size_t n = 2 * 64 * 2 * 64 + 16 * 2 * 64;
    hipLaunchKernelGGL(chain_pack_kernel, dim3((unsigned)sdn::div_up<size_t>(n, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_conv_chain_pack_weights");
}

int sdn_conv_chain(const void *in_hi, const void *in_lo, const void *packed, const float *consts, float *out_img, float *out_raw,
                   int H, int W, int n_workgroups, sdn_stream_t stream) {
    SDN_REQUIRE(in_hi && in_lo && packed && consts && out_img && H > 0 && W > 0, "sdn_conv_chain: bad argument");
    ChainParams p;
    p.yh = (const _Float16 *)in_hi; p.yl = (const _Float16 *)in_lo; p.wpk = (const half8 *)packed; p.consts = consts; p.img = out_img;
    p.raw = out_raw;
    p.H = H; p.W = W;
    int Hb, Wb;
    sdn_conv_plane_dims(H, W, &Hb, &Wb);
    p.Wb = Wb;
    p.chunk_bytes = (long)Hb * Wb * 32;
    p.tiles_per_row = sdn::div_up(W, 32);
    p.n_tiles = p.tiles_per_row * H;
    const int n_groups = sdn::div_up(p.n_tiles, 4);
    int wg = n_workgroups > 0 ? n_workgroups : 256;
    if (wg > n_groups) wg = n_groups;
    hipLaunchKernelGGL(chain_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_conv_chain");
}

size_t sdn_conv_head_packed_weight_bytes(void) { return HEAD_FRAGS * sizeof(half8); }

int sdn_conv_head_pack_weights(const float *w1, void *packed, sdn_stream_t stream) {
    SDN_REQUIRE(w1 && packed, "sdn_conv_head_pack_weights: null pointer");
    HeadPackParams p;
    p.w1 = w1; p.out = (half8 *)packed;
    hipLaunchKernelGGL(head_pack_kernel, dim3(sdn::div_up(HEAD_UNITS * 2 * 64, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_conv_head_pack_weights");
}

int sdn_conv_head(const float *x, const void *packed, const float *bias, void *out_hi, void *out_lo, int H, int W, int n_workgroups,
                  sdn_stream_t stream) {
    SDN_REQUIRE(x && packed && bias && out_hi && out_lo && H > 0 && W > 0, "sdn_conv_head: bad argument");
    HeadParams p;
    p.x = x; p.wpk = (const half8 *)packed; p.bias = bias; p.oh = (_Float16 *)out_hi; p.ol = (_Float16 *)out_lo;
    p.H = H; p.W = W;
    int Hb, Wb;
    sdn_conv_plane_dims(H, W, &Hb, &Wb);
    p.Wb = Wb;
    p.chunk_elems = (long)Hb * Wb * 16;
    p.tiles_per_row = sdn::div_up(W, 32);
    p.n_tiles = p.tiles_per_row * H;
    const int n_groups = sdn::div_up(p.n_tiles, 4);
    int wg = n_workgroups > 0 ? n_workgroups : 256;
    if (wg > n_groups) wg = n_groups;
    hipLaunchKernelGGL(head_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_conv_head");
}

size_t sdn_sky_packed_weight_bytes(void) { return SKY_PACKED_FRAGS * sizeof(half8); }
size_t sdn_sky_consts_floats(void) { return SC_TOTAL; }

int sdn_sky_pack_weights(const float *w1, const float *const *wh4_host, const float *wc, void *packed, sdn_stream_t stream) {
    SDN_REQUIRE(w1 && wh4_host && wc && packed, "sdn_sky_pack_weights: null pointer");
    SkyPackParams p;
    p.w1 = w1;
    for (int i = 0; i < 4; i++) {
        SDN_REQUIRE(wh4_host[i], "sdn_sky_pack_weights: null hidden weight");
        p.wh[i] = wh4_host[i];
    }
    p.wc = wc;
    p.out = (half8 *)packed;
    const size_t n = 16 * 2 * 64 + 4 * 64 * 2 * 64 + 16 * 2 * 64;
    hipLaunchKernelGGL(sky_pack_kernel, dim3((unsigned)sdn::div_up<size_t>(n, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_sky_pack_weights");
}

int sdn_sky_pack_weights_mx(const float *w1, const float *const *wh4_host, const float *wc, void *packed, sdn_stream_t stream) {
    if (int rc = sdn_sky_pack_weights(w1, wh4_host, wc, packed, stream)) return rc;
    PackMxParams p;
    for (int i = 0; i < 4; i++) p.wh[i] = wh4_host[i];
    p.n_layers = 4;
    p.base = SKY_L0_FRAGS;
    p.out = (half8 *)packed;
    hipLaunchKernelGGL(pack_mx_kernel, dim3(4 * 64 * 64 / 256), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_sky_pack_weights_mx");
}

static int sky_workgroups(int32_t n_rays, int32_t n_workgroups) {
    int wg = n_workgroups > 0 ? n_workgroups : 256;
    const int groups = sdn::div_up(sdn::div_up(n_rays, 32), 4);
    return wg > groups ? groups : wg;
}

int32_t sdn_sky_partial_rows(int32_t n_rays, int32_t n_workgroups) { return n_rays > 0 ? 4 * sky_workgroups(n_rays, n_workgroups) : 0; }

int sdn_sky_mlp(const float *raydirs, const void *packed, const float *consts, float *sky_c, float *sky_partial, int32_t n_rays,
                int32_t n_workgroups, float *sky_avg, uint32_t *counter, int32_t hidden_terms, int32_t encoded, sdn_stream_t stream) {
    SDN_REQUIRE(raydirs && packed && consts && sky_c && sky_partial && n_rays > 0, "sdn_sky_mlp: bad argument");
    SDN_REQUIRE((sky_avg == nullptr) == (counter == nullptr), "sdn_sky_mlp: sky_avg and counter go together");
    SkyParams p;
    p.sky_avg = sky_avg; p.counter = counter;
    p.raydirs = raydirs; p.wpk = (const half8 *)packed; p.consts = consts; p.sky_c = sky_c; p.sky_partial = sky_partial;
    p.R = n_rays;
    p.n_tiles = sdn::div_up(n_rays, 32);
    const int wg = sky_workgroups(n_rays, n_workgroups);
    SDN_REQUIRE(hidden_terms == 3 || hidden_terms == 6, "sdn_sky_mlp: hidden_terms must be 3 or 6");
    SDN_REQUIRE(encoded == 0 || encoded == 1, "sdn_sky_mlp: encoded must be 0 (ray directions) or 1 (positional-encoded rows)");
    if (encoded) {
        SDN_REQUIRE(hidden_terms == 3, "sdn_sky_mlp: positional-encoded input rows are evaluated with the 3-term split only");
        hipLaunchKernelGGL((sky_kernel<0, 0, true>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    } else if (hidden_terms == 6) hipLaunchKernelGGL((sky_kernel<0, 1>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((sky_kernel<0, 0>), dim3(wg), dim3(256), 0, (hipStream_t)stream, p);
    return sdn::check_launch("sdn_sky_mlp");
}

int sdn_debug_mfma_probe(const float *A, const float *B, float *C, sdn_stream_t stream) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C);
    return sdn::check_launch("sdn_debug_mfma_probe");
}

}  // extern "C"
