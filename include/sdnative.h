/*
 * sdnative.h -- C ABI of libsdnative.so: the MI355X (gfx950) implementation of
 * SceneDreamer's inference hot path.
 *
 * Conventions
 *   - every pointer marked "dev" is a device (HBM) pointer owned by the caller
 *     (PyTorch's caching allocator in practice); nothing is allocated, retained
 *     or freed by the library across calls unless a function says otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = legacy default stream);
 *     every entry point only ENQUEUES work on it and returns immediately;
 *   - return value: 0 on success, negative sdn_status on failure; the message
 *     of the last failure on the calling thread is sdn_last_error();
 *   - nothing throws across this boundary.
 *
 * Each entry point names the reference interface (file:line under
 * FrozenBurning/SceneDreamer) it replaces.  The Python modules in
 * scenedreamer_amd/shims/ (`voxlib`, `_gridencoder`, `gridencoder`) bind these
 * symbols with ctypes and re-create the reference's pybind signatures.
 */
#ifndef SDNATIVE_H
#define SDNATIVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5: window_host grew from int32[5] to int32[6] (element 5: 8 x 4 pixel-block ray order) and sdn_field_aux gained
 *    `colour_passes` + `flags` -- a caller built against version 4 would be read past its arrays / struct. */
#define SDN_ABI_VERSION 5

typedef void *sdn_stream_t; /* hipStream_t */

enum sdn_status {
    SDN_OK = 0,
    SDN_ERR_INVALID = -1,     /* bad argument (null pointer, bad size)          */
    SDN_ERR_UNSUPPORTED = -2, /* D/C/dtype combination the reference also rejects */
    SDN_ERR_LAUNCH = -3       /* HIP reported a launch/runtime error             */
};

enum sdn_dtype { SDN_F32 = 0, SDN_F16 = 1 };

int sdn_abi_version(void);
const char *sdn_last_error(void);

/* ---------------------------------------------------------------------------
 * voxlib.ray_voxel_intersection_perspective
 *   replaces ray_voxel_intersection_perspective_cuda
 *   (imaginaire/model_utils/gancraft/voxlib/ray_voxel_intersection.cu:253-325,
 *    binding voxlib.cpp:26).
 *   vox        dev int32, element (x,y,z) at vox[x*strides[0]+y*strides[1]+z*strides[2]]
 *   dims       host int64[3] voxel grid extent; strides host int64[3] in elements
 *   cam_ori / cam_dir / cam_up   host float[3]; cam_c host float[2]; img_dims host int[2] = {H, W}
 *   out_voxel_id  dev int32 [H, W, max_samples, 1]
 *   out_depth     dev f32   [2, H, W, max_samples, 1]   (entry t, exit t2; NaN on miss)
 *   out_raydirs   dev f32   [H, W, 1, 3]
 *   occupancy  dev u8 block-occupancy grid of THIS volume from sdn_rvip_build_occupancy, or NULL.  With it, rays
 *              cross empty 8x16x16-cell blocks in one step; the outputs are the same bits either way.
 * Results are bit-identical to the reference source evaluated without FMA
 * contraction (see oracle/sdn_oracle.c).
 */
int sdn_rvip(const int32_t *vox, const int64_t *dims, const int64_t *strides, const float *cam_ori,
             const float *cam_dir, const float *cam_up, float cam_f, const float *cam_c, const int *img_dims,
             int max_samples, const uint8_t *occupancy, int32_t *out_voxel_id, float *out_depth, float *out_raydirs,
             sdn_stream_t stream);
/* acceleration structure for sdn_rvip (no reference counterpart; rebuild whenever the volume changes):
 * bytes needed for a volume of extent dims (0 for bad dims), and the build (one pass over the volume). */
size_t sdn_rvip_occupancy_bytes(const int64_t *dims);
int sdn_rvip_build_occupancy(const int32_t *vox, const int64_t *dims, const int64_t *strides, uint8_t *occupancy,
                             sdn_stream_t stream);

/* Compact volume (no reference counterpart): the same ray marcher over a uint8 volume of PALETTE INDICES
 * (0 = empty) with palette256 dev int32[256] (palette256[0] == 0) -- a world holds ~20 distinct block ids, so the
 * volume is 4x smaller than the reference's int32 ids (PCGVoxelGenerator.voxel_t, pcg_gen.py:119,173) while
 * out_voxel_id carries the same int32 block ids: outputs are bit-identical to sdn_rvip on the expanded volume. */
int sdn_rvip_u8(const uint8_t *vox, const int32_t *palette256, const int64_t *dims, const int64_t *strides,
                const float *cam_ori, const float *cam_dir, const float *cam_up, float cam_f, const float *cam_c,
                const int *img_dims, int max_samples, const uint8_t *occupancy, int32_t *out_voxel_id, float *out_depth,
                float *out_raydirs, sdn_stream_t stream);
int sdn_rvip_build_occupancy_u8(const uint8_t *vox, const int64_t *dims, const int64_t *strides, uint8_t *occupancy,
                                sdn_stream_t stream);
/* Measurement entry point (no reference counterpart; SURVEY 8(d): the ray marcher's algorithmic bytes are 4 B x DDA steps +
 * 84 B per ray): the launch of sdn_rvip (palette256 == NULL, vox int32) or sdn_rvip_u8 (palette256 != NULL, vox uint8) with the
 * walk's counters summed over all rays into counters dev u64[4], ZEROED by the caller: [0] loop iterations, [1] volume reads,
 * [2] empty-block jumps, [3] rays.  With occupancy == NULL, [0] is the number of cell-by-cell DDA steps the reference's loop
 * executes for this frame (ray_voxel_intersection.cu:115-229).  Outputs are written as by sdn_rvip. */
int sdn_rvip_debug_counts(const void *vox, const int32_t *palette256, const int64_t *dims, const int64_t *strides,
                          const float *cam_ori, const float *cam_dir, const float *cam_up, float cam_f, const float *cam_c,
                          const int *img_dims, int max_samples, const uint8_t *occupancy, int32_t *out_voxel_id, float *out_depth,
                          float *out_raydirs, uint64_t *counters, sdn_stream_t stream);
/* int32 ids (any strides) -> palette indices, out dev u8 [dims] contiguous.  id2idx dev u8[n_ids]: index of every id
 * (id2idx[0] == 0); *bad_flag (dev int32, zeroed by the caller) is set when the volume holds an id outside the table
 * or one the palette does not contain. */
int sdn_volume_compact(const int32_t *vox, const int64_t *dims, const int64_t *strides, const uint8_t *id2idx, int n_ids,
                       uint8_t *out, int32_t *bad_flag, sdn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Scene ingestion on the GPU: PCGVoxelGenerator.next_world (imaginaire/model_utils/pcg_gen.py:119-164) writing the
 * compact volume directly.  volume dev u8 [sample_height, S0, S1], zeroed by the caller.
 *   sdn_scene_columns      terrain shell (:121-128): cells h .. h+pad_num of every column (clipped to the volume) =
 *                          column_idx (dev u8 [S0,S1], palette index of the column's biome block); height_map dev i16.
 *   sdn_scene_paste_trees  (:134-159) trees_hxym dev int32 [n_trees,4] = (h, x, y, model); models dev u8 (palette
 *                          indices, 0 = empty) concatenated, model m at model_offsets[m] with extent model_dims[3m..];
 *                          a model cell is written only where the world is still empty.  Trees of ONE call must not
 *                          overlap each other: the caller issues overlapping trees in successive calls, in tree order
 *                          (an earlier tree keeps the cell, like the reference's sequential loop).
 *   sdn_scene_column_tops  (:162-164) top dev int32 [S0,S1]: highest non-empty cell of every column, 0 if none. */
int sdn_scene_columns(const int16_t *height_map, const uint8_t *column_idx, int sample_height, int S0, int S1, int pad_num,
                      uint8_t *volume, sdn_stream_t stream);
int sdn_scene_paste_trees(uint8_t *volume, int sample_height, int S0, int S1, const int32_t *trees_hxym, int n_trees,
                          const uint8_t *models, const int32_t *model_offsets, const int32_t *model_dims, sdn_stream_t stream);
int sdn_scene_column_tops(const uint8_t *volume, int sample_height, int S0, int S1, int32_t *top, sdn_stream_t stream);

/* ---------------------------------------------------------------------------
 * voxlib.positional_encoding / positional_encoding_backward
 *   replaces positional_encoding_cuda / positional_encoding_backward_cuda
 *   (.../voxlib/positional_encoding_kernel.cu:129-197, :209-285; voxlib.cpp:29-30).
 *   in  dev f32 [pre, post]  ->  out dev f32 [pre, 2*ndegrees(+1), post]
 *   (pre = product of dims before `dim`, post = product of dims from `dim` on)
 */
int sdn_posenc_fwd(const float *in, float *out, int64_t pre, int64_t post, int ndegrees, int incl_orig,
                   sdn_stream_t stream);
int sdn_posenc_bwd(const float *out_grad, const float *out, float *in_grad, int64_t pre, int64_t post,
                   int ndegrees, int incl_orig, sdn_stream_t stream);

/* ---------------------------------------------------------------------------
 * _gridencoder.grid_encode_forward / grid_encode_backward
 *   replaces gridencoder/src/gridencoder.cu:423-446 / :448-478 (gridencoder.h:12-13).
 *   inputs      dev f32 [B, D] in [0,1]
 *   embeddings  dev [sO, C] of emb_dtype (SDN_F32 | SDN_F16)
 *   offsets     dev int32 [L+1]
 *   outputs     dev [L, B, C] of emb_dtype, written in place
 *   dy_dx       dev [B, L*D*C] of emb_dtype when calc_grad_inputs, else ignored
 *   S = log2(per_level_scale), H = base resolution, gridtype 0 hash / 1 tiled
 * D in {2,3,4,5}, C in {1,2,4,8}; anything else returns SDN_ERR_UNSUPPORTED,
 * as the reference throws (gridencoder.cu:355,372).
 */
int sdn_grid_encode_fwd(const float *inputs, const void *embeddings, int emb_dtype, const int32_t *offsets,
                        void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                        int calc_grad_inputs, void *dy_dx, uint32_t gridtype, int align_corners,
                        sdn_stream_t stream);
/*   per-level scale = exp2f(l*S)*H-1 and resolution = ceil(scale)+1 exactly as every kernel of this library
 *   uses them (gridencoder.cu:126-127), evaluated on the host; resolutions_host may be NULL              */
int sdn_grid_level_scales(uint32_t L, float S, uint32_t H, float *scales_host, uint32_t *resolutions_host);
/*   grad [L,B,C]; grad_embeddings [sO,C] pre-zeroed by the caller (accumulated with atomics);
 *   grad_inputs [B,D] written when calc_grad_inputs.  emb_dtype SDN_F32 or SDN_F16 (grad and grad_embeddings in that type:
 *   __half2 atomics for f16, as gridencoder.cu:227-343 does).                                                            */
int sdn_grid_encode_bwd(const void *grad, const float *inputs, const void *embeddings, int emb_dtype,
                        const int32_t *offsets, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                        uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void *dy_dx,
                        void *grad_inputs, uint32_t gridtype, int align_corners, sdn_stream_t stream);


/* ---------------------------------------------------------------------------
 * Fused field renderer: fast path for Generator._forward_perpix / _forward_perpix_sub
 * (imaginaire/generators/scenedreamer.py:285-430) = mc_utils.sample_depth_batched (:82-151) +
 * label lookup (scenedreamer.py:357-363) + GridEncoder.forward (gridencoder/grid.py:140-156) +
 * LightningMLP.forward (imaginaire/model_utils/layers.py:92-126) + volum_rendering_relu
 * (mc_utils.py:154-161) + sky compositing (scenedreamer.py:373-413), inference settings
 * (deterministic sampling, keep_sky_out_avgpool, clip_feat_map=True, N=1).
 * Only the SceneDreamer grid (D=5, C=8, 16 hashed power-of-two levels) is supported here; anything else
 * returns SDN_ERR_UNSUPPORTED and callers use the three drop-in ops above.
 *
 * Call order: collapse_table once per scene (global_enc), pack_weights once per style code, then per
 * frame encode -> mlp.  All scratch buffers are caller-allocated (sizes from the *_bytes/_elems helpers).
 */
size_t sdn_field_packed_weight_bytes(void);
size_t sdn_field_consts_floats(void);
/* float offset of a block inside `consts`: 0 label_bias[12][256] (= fc_m_a^T + fc_1.bias), 1 beta[5][256],
 * 2 w_sigma[256], 3 b_c[64], 4 b_sigma[1], 5 sky_avg[64] (updated by the caller every frame) */
int sdn_field_const_offset(int which);
size_t sdn_field_feat_bytes(int32_t n_rays, int32_t num_samples);
size_t sdn_field_aux_elems(int32_t n_rays, int32_t num_samples);

/* embeddings dev f32 [sO,8]; offsets_host int32[L+1]; genc_host f32[2] = world_encoder output;
 * table3 dev f32 [16, T, 8] (T = rows per level) */
int sdn_field_collapse_table(const float *embeddings, const int32_t *offsets_host, uint32_t L, float S, uint32_t H,
                             const float *genc_host, float *table3, sdn_stream_t stream);
/* w1 dev [256,128] fc_1.weight; wh5_host: host array of 5 dev pointers [256,256] = fc_2..fc_6 weight * alpha;
 * wc dev [64,256] fc_out_c.weight; packed dev, sdn_field_packed_weight_bytes() bytes.
 * The trunk layers fc_1 .. fc_4 are stored times 2^sdn_field_trunk_shift() (the MLP kernels take the factor back out), so
 * that the lo halves of the f16 split leave f16's subnormal range: the caller must keep
 * max(|fc_1.weight|, 0.4 |fc_2..4 weight * alpha|) * 2^sdn_field_trunk_shift() below f16's 65504 (not checked here: the
 * weights are device memory; the host wrapper fused.prepare_style checks it). */
int sdn_field_trunk_shift(void);
int sdn_field_pack_weights(const float *w1, const float *const *wh5_host, const float *wc, void *packed,
                           sdn_stream_t stream);
/* The same stream (same size) with the colour layers fc_5 / fc_6 laid out for colour_terms = 6 of sdn_field_mlp: f16 Whi
 * fragments + block-scaled fp6 (e2m3) fragments of Wlo and Whi with one E8M0 scale per row and 32-k block. */
int sdn_field_pack_weights_mx(const float *w1, const float *const *wh5_host, const float *wc, void *packed,
                              sdn_stream_t stream);
/* voxel_id dev i32 [R,M]; depth2 dev f32 [2,R,M]; raydirs dev f32 [R,3]; lut1024 dev u8 [1024] block id ->
 * reduced label (ignore already mapped to dirt); scales_dev f32 [16]; outputs: feat, dist, label (aux_elems each),
 * rayflag u8 [R].  Sample placement = mc_utils.sample_depth_batched(nsamples = num_samples + 1, use_box_boundaries =
 * False): deterministic (inference) with u_dev = NULL and lin_dev f32 [num_samples+1] = linspace(0,1,num_samples+3)[1:-1];
 * stochastic / stratified (training, mc_utils.py:121-125) with u_dev f32 [R, num_samples+1] = the caller's torch.rand draw
 * and lin_dev = linspace(0,1,num_samples+2)[:-1]; strat_division selects how `rand_samples / nsamples` (mc_utils.py:123,
 * tensor / Python scalar) is evaluated: 0 = multiplication by the float32 reciprocal, what PyTorch does on a CUDA tensor
 * (the reference's GPU path); 1 = IEEE division, what PyTorch does on a CPU tensor (the goldens recorded from the
 * reference's CPU run).  They differ by at most 1 ulp, and only when nsamples is not a power of two.
 * window_host: NULL (the n_rays rays are rays 0..n_rays-1 of voxel_id / depth2 / raydirs), or host int32[6]
 *   {n_src, pitch, first, cols, ray0, blocked}: the arrays hold n_src rays (the whole padded frame the ray marcher wrote, as
 *   the reference's per-frame voxlib call does, scenedreamer.py:576-590) and local ray r is ray w = ray0 + r of a window of
 *   `cols` columns whose ray (y, x) is source ray first + y * pitch + x (cols = 0: source ray = ray0 + r).  Outputs
 *   (feat, dist, label, rayflag) are indexed by the LOCAL ray.
 *   blocked != 0 (needs ray0 = 0, cols = 8k, n_rays = cols x 4m: the launch is the whole window): the launch's ray ORDER is 8 x 4
 *   pixel blocks instead of row-major -- the 32 rays a workgroup takes through a pass together are neighbours in both directions,
 *   so a group's all-or-nothing savings (colour-branch skipping, early termination) apply more often; per-ray results are the
 *   same bits, the per-ray outputs and inputs that are not read through the window (net_out, the sdn_field_aux arrays, u_dev) stay
 *   in the window's row-major order; `passes` / `colour_passes` then count per block.  sdn_field_encode and sdn_field_mlp of one
 *   frame must be given the same value.
 *   blocked == 2 (sdn_field_render only; ray0 = 0, n_rays = rows x cols, ANY rows / cols): the same block order over the
 *   ceil(cols / 8) x ceil(rows / 4) block grid that covers the window; block positions outside the window are no rays.  `passes` /
 *   `colour_passes` then hold ceil(cols / 8) * ceil(rows / 4) entries (>= ceil(n_rays / 32)); u_dev must be NULL.  With whole
 *   blocks it is the same launch as blocked == 1. */
int sdn_field_encode(const int32_t *voxel_id, const float *depth2, const float *raydirs, const uint8_t *lut1024,
                     const float *table3, uint32_t table_rows, const float *scales_dev, const float *genc_host,
                     const float *cam_ori_host, const float *voxel_dims_host, const float *lin_dev, const float *u_dev,
                     int32_t n_rays, int32_t max_blocks, int32_t num_samples, float sample_depth, float dists_scale, float *feat,
                     float *dist, uint8_t *label, uint8_t *rayflag, const int32_t *window_host, int32_t strat_division,
                     sdn_stream_t stream);
/* mc_utils.sample_depth_batched (imaginaire/model_utils/gancraft/mc_utils.py:82-151, use_box_boundaries = False) as an op:
 * depth2 dev f32 [2,R,M] -> rand_depth, new_dists dev f32 [R, n_points-1], idx dev i64 [R, n_points-1] (raw values: NaN
 * depths of rays without a hit are left for the caller to zero, scenedreamer.py:350-352).  lin_dev / u_dev as above with
 * n_points = nsamples of the reference call, strat_division as for sdn_field_encode. */
int sdn_sample_depth(const float *depth2, const float *lin_dev, const float *u_dev, int32_t n_rays, int32_t max_blocks,
                     int32_t n_points, float sample_depth, float *rand_depth, float *new_dists, int64_t *idx,
                     int32_t strat_division, sdn_stream_t stream);
/* sky_c dev f32 [R,64] = sky_net output per ray; net_out dev f32 [R,64]; n_workgroups <= 0 -> one per CU.
 * colour_terms: products of the colour layers fc_5 / fc_6 (LightningMLP.forward, layers.py:117-124): 3 = the 3-term f16 split
 *   like every other layer; 2 = without Whi.Xlo; 6 = Whi.Xhi in f16 + the two correction terms as block-scaled fp6
 *   products at 4x the f16 MFMA rate (`packed` must then come from sdn_field_pack_weights_mx).
 * term_eps: early ray termination -- a 32-ray group stops sampling once the transmittance of all its rays is below
 *   term_eps (changes net_out by at most 2 * term_eps); 0 = off (the reference evaluates every sample).
 * passes: optional dev u8 [ceil(ceil(R / 8) / 4)], number of 4-sample passes every 32-ray group went through.
 * window_host: as for sdn_field_encode; it applies to sky_c (indexed by the SOURCE ray: the sky MLP covers the whole
 *   padded frame), net_out is indexed by the local ray.
 * sky_avg: NULL (the value at const offset 5 is used) or dev f32 [64], the frame mean sdn_sky_mlp finished.
 * ticket: NULL (workgroup b of G evaluates the 32-ray groups b, b + G, ...; groups whose rays all miss are skipped) or dev
 *   int32[2], ZERO before the first launch and left zero by every launch: after two static rounds the persistent
 *   workgroups draw their groups from this counter, so none idles at the end because its share was mostly sky.  One
 *   launch at a time per ticket buffer.  net_out does not depend on the schedule. */
int sdn_field_mlp(const float *feat, const float *dist, const uint8_t *label, const uint8_t *rayflag, const void *packed,
                  const float *consts, const float *sky_c, float *net_out, int32_t n_rays, int32_t num_samples,
                  int32_t colour_terms, float term_eps, uint8_t *passes, int32_t n_workgroups, const int32_t *window_host,
                  const float *sky_avg, int32_t *ticket, sdn_stream_t stream);
/* sdn_field_encode + sdn_field_mlp as ONE kernel (the north star's "trilinear hash-grid lookup plus the tiny sigma/color MLP
 * fused into one kernel"; replaces Generator._forward_perpix's hash_encoder -> render_net call chain, scenedreamer.py:298-311,
 * with the sample placement in front of it, :341-363, and the compositing behind it, :373-413): every 4-sample pass of a
 * 32-ray group starts with its own sample placement and collapsed-table gathers, executed by the wave that then runs the
 * MLP on them, straight into its MFMA operand registers.  No feature / dist / label / rayflag buffers; net_out is the same
 * bits as the two-kernel sequence produces.  Arguments: those of sdn_field_encode (inputs) and of sdn_field_mlp (weights,
 * sky, outputs, schedule) with the same meaning; colour_terms 3 or 6.
 *   cam_ori_dev: NULL, or dev f32 [3] -- the camera origin read from device memory instead of cam_ori_host (which may then be
 *     NULL): Generator._forward_perpix receives it as the device tensor cam_ori_t (scenedreamer.py:313, :354).
 *   aux: NULL, or a host struct of optional device pointers (term_eps must be 0 when any is set) that receive the OTHER return
 *     values of _forward_perpix (scenedreamer.py:429-430) for the rays of this launch (local ray index), so that a binding of
 *     that method can return its whole 12-tuple: weights (:373-376; consumed by inference_givenstyle_depth, :812-817), depth =
 *     rand_depth after the NaN / inf -> 0 replacement (:346-352), sigma = net_out_s and colour = net_out_c (LightningMLP's two
 *     outputs, layers.py:114, :124), sky_blended = skynet_out_c after the keep_sky_out blend (:401), nosky = nosky_mask (:382).
 *     With aux set no 32-ray group is skipped and rays that hit nothing are gathered too (the reference evaluates them). */
typedef struct sdn_field_aux {
    float *weights;      /* dev f32 [n_rays, num_samples]     */
    float *depth;        /* dev f32 [n_rays, num_samples]     */
    float *sigma;        /* dev f32 [n_rays, num_samples]     */
    float *colour;       /* dev f32 [n_rays, num_samples, 64] */
    float *sky_blended;  /* dev f32 [n_rays, 64]              */
    uint8_t *nosky;      /* dev u8  [n_rays]                  */
    /* the two below are diagnostics and do NOT switch the launch to the per-sample-output form: */
    uint8_t *colour_passes; /* dev u8 [ceil(n_rays / 32)]: how many passes of each 32-ray group evaluated the colour branch */
    int32_t flags;          /* SDN_FIELD_* bits */
} sdn_field_aux;
/* Colour-branch skipping: a pass (4 samples of each of a workgroup's 32 rays) whose 128 samples ALL have relu(sigma) * dist == 0
 * -- volume-rendering weight exactly 0, mc_utils.py:154-161 -- does not evaluate fc_5 / fc_6 / fc_out_c: the colours would be
 * multiplied by zero, net_out is bit-identical.  On by default; this flag evaluates every pass in full (A/B timing, tests). */
#define SDN_FIELD_NO_COLOUR_SKIP 1
int sdn_field_render(const int32_t *voxel_id, const float *depth2, const float *raydirs, const uint8_t *lut1024,
                     const float *table3, uint32_t table_rows, const float *scales_dev, const float *genc_host,
                     const float *cam_ori_host, const float *voxel_dims_host, const float *lin_dev, const float *u_dev,
                     int32_t n_rays, int32_t max_blocks, int32_t num_samples, float sample_depth, float dists_scale,
                     const void *packed, const float *consts, const float *sky_c, const float *sky_avg, float *net_out,
                     int32_t colour_terms, float term_eps, uint8_t *passes, int32_t n_workgroups, const int32_t *window_host,
                     int32_t strat_division, int32_t *ticket, const float *cam_ori_dev, const sdn_field_aux *aux, sdn_stream_t stream);

/* LightningMLP.forward as an op (imaginaire/model_utils/layers.py:92-126; the nn.Module boundary of SURVEY 8(b)) for N = 1:
 *   x dev f32 [n_rows, 128] hash-grid features; label dev u8 [n_rows] = index of the one in each row of the one-hot mask `m`
 *   (fc_m_a(m) is a row lookup then); packed / consts as for sdn_field_render (ModLinear's per-style W * alpha and beta folded:
 *   layers.py:247-269 with N = 1); outputs sigma dev f32 [n_rows] (fc_sigma, :114) and c dev f32 [n_rows, 64] (fc_out_c, :124).
 *   The same MFMA layer machinery and arithmetic (3-term f16 split; colour_terms 3 or 6) as the fused field kernel.
 *   ticket: as for sdn_field_mlp (NULL = static schedule). */
int sdn_render_mlp(const float *x, const uint8_t *label, const void *packed, const float *consts, float *sigma, float *c,
                   int64_t n_rows, int32_t colour_terms, int32_t n_workgroups, int32_t *ticket, sdn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Render CNN: the convolutions of RenderCNN (imaginaire/generators/gancraft_base.py:175-225, forward :202-225) on MFMA
 * as f16 products with f32 accumulation: 3x3 256->256 (conv2a/2b/3a/3b; taps = 9, cin = 256) and 1x1 cin->256
 * (conv1 64->256, conv4a/4b; taps = 1); the final conv4 (256->3, 1x1) + tanh (:221, :603) is an optional projection
 * in the epilogue.
 * terms = 3: every product is Whi.Xhi + Wlo.Xhi + Whi.Xlo (the field MLP's split, ~2^-21 relative);
 * terms = 1: Whi.Xhi only, hi = round-to-nearest f16 (3x3 layers only; in_lo may be NULL).  The weights must have been
 * packed with the same `terms`.
 * Activations travel between the convolutions as two f16 planes (hi, lo) [cin/16 chunks][Hb*Wb][16 channels] with a zero
 * border (extent from sdn_conv_plane_dims; the caller zero-fills the planes ONCE, kernels never write the border or
 * pixels outside the H x W frame); fp32 tensors are rows [H*W][256] (channels last).
 *   y   = LeakyReLU_0.2( (resid + conv(in) + bias) * (mod_w + 1) + mod_b )       each term optional; the residual is
 *         given either as fp32 rows (resid) or as hi/lo planes (resid_hi/lo, which MAY be the output planes: in place)
 *   img = tanh(proj_w . y + proj_b)                                                optional, [3][H*W]
 */
void sdn_conv_plane_dims(int H, int W, int *Hb, int *Wb);
/* 0 for an unsupported (cin, taps, terms) */
size_t sdn_conv_packed_weight_bytes(int cin, int taps, int terms);
/* w_oihw dev f32 [256,cin,k,k] (k*k = taps) -> packed dev */
int sdn_conv_pack_weights(const float *w_oihw, int cin, int taps, int terms, void *packed, sdn_stream_t stream);
/* x dev f32 [H*W,channels] -> hi/lo planes */
int sdn_conv_planes_from_f32(const float *x, int channels, void *out_hi, void *out_lo, int H, int W, sdn_stream_t stream);
/* outputs: any of out_hi (+ out_lo; NULL when every consumer is 1-term) planes, out_f32 rows [H*W,256], out_img [3,H*W]
 * (with proj_w [3,256], proj_b [3]) */
int sdn_conv(const void *in_hi, const void *in_lo, int cin, int taps, int terms, const void *packed, const float *bias,
             const float *resid, const void *resid_hi, const void *resid_lo, const float *mod_w, const float *mod_b,
             void *out_hi, void *out_lo, float *out_f32, const float *proj_w, const float *proj_b, float *out_img, int H,
             int W, int n_workgroups, sdn_stream_t stream);

/* The head of RenderCNN.forward as ONE kernel: y = LeakyReLU_0.2(conv1(x) + bias) (gancraft_base.py:206), x dev f32 rows
 * [H*W][64] (net_out), y as f16 hi / lo planes -- the same result as sdn_conv_planes_from_f32 -> sdn_conv(conv1) to f32
 * rounding, without the 64-channel planes in between.  w1 dev f32 [256,64]; bias dev f32 [256]; the output planes are
 * zero-filled by the caller once (only pixels of the H x W frame are written). */
size_t sdn_conv_head_packed_weight_bytes(void);
int sdn_conv_head_pack_weights(const float *w1, void *packed, sdn_stream_t stream);
int sdn_conv_head(const float *x, const void *packed, const float *bias, void *out_hi, void *out_lo, int H, int W, int n_workgroups,
                  sdn_stream_t stream);

/* The tail of RenderCNN.forward as ONE kernel: img = tanh(conv4(LeakyReLU(y + conv4b(LeakyReLU(conv4a(y))))))
 * (imaginaire/generators/gancraft_base.py:219-225, tanh :603) -- the per-pixel 256 -> 256 -> 256 -> 3 chain evaluated
 * register-resident on the field MLP's layer machinery (3-term f16 split everywhere): the 256-channel activation is read
 * ONCE and never written.  Same result as  sdn_conv(conv4a) -> sdn_conv(conv4b, resid = y, proj = conv4)  to f32 rounding.
 *   in_hi / in_lo: y as f16 planes (the layout of sdn_conv);  out_img dev f32 [3][H*W];  out_raw: NULL or dev f32 [3][H*W], conv4's
 *   output BEFORE tanh -- RenderCNN.forward's own return value (gancraft_base.py:221-225; _forward_global returns both, :598-603)
 *   packed: sdn_conv_chain_packed_weight_bytes() bytes from sdn_conv_chain_pack_weights (w4a, w4b dev f32 [256,256]; w4 dev f32 [3,256])
 *   consts dev f32 [sdn_conv_chain_consts_floats()]: conv4a.bias[256] | conv4b.bias[256] | conv4.bias padded with zeros to 64 */
size_t sdn_conv_chain_packed_weight_bytes(void);
size_t sdn_conv_chain_consts_floats(void);
int sdn_conv_chain_pack_weights(const float *w4a, const float *w4b, const float *w4, void *packed, sdn_stream_t stream);
int sdn_conv_chain(const void *in_hi, const void *in_lo, const void *packed, const float *consts, float *out_img, float *out_raw,
                   int H, int W, int n_workgroups, sdn_stream_t stream);

/* Sky MLP for every ray + per-feature sum over rays: SKYMLP.forward(positional_encoding(raydirs, 5, incl_orig), z)
 * (imaginaire/generators/gancraft_base.py:150-169; the frame mean of scenedreamer.py:592-598 = column sums of sky_partial / n_rays).
 * consts (sdn_sky_consts_floats floats): [fc1.bias + fc_z_a(z) : 256][fc2..fc5 bias : 4x256][fc_out_c.bias : 64];
 * w1 dev [256,33]; wh4_host host array of 4 dev pointers [256,256]; wc dev [64,256];
 * sky_partial dev f32 [sdn_sky_partial_rows(n_rays, n_workgroups), 64]: every wave's sum of its rays' sky_c (all rows are
 * written -- no float atomics, the mean is reproducible bit for bit).
 * sky_avg + counter (both or neither): dev f32 [64] and a dev uint32 that is ZERO before the first launch; the last
 * workgroup to finish adds the partial rows in row order (double accumulation) and writes the frame mean sum / n_rays,
 * then resets the counter.  Without them the caller adds the rows up. */
int32_t sdn_sky_partial_rows(int32_t n_rays, int32_t n_workgroups);
size_t sdn_sky_packed_weight_bytes(void);
size_t sdn_sky_consts_floats(void);
int sdn_sky_pack_weights(const float *w1, const float *const *wh4_host, const float *wc, void *packed, sdn_stream_t stream);
/* the same stream with the hidden layers fc2..fc5 laid out for hidden_terms = 6 (f16 Whi fragments + block-scaled fp6) */
int sdn_sky_pack_weights_mx(const float *w1, const float *const *wh4_host, const float *wc, void *packed, sdn_stream_t stream);
/* hidden_terms: products of fc2..fc5: 3 = the 3-term f16 split, 6 = Whi.Xhi in f16 + block-scaled fp6 corrections (packed from
 * sdn_sky_pack_weights_mx); fc1 and fc_out_c always use the 3-term split.
 * encoded: 0 = `raydirs` dev f32 [n_rays,3], the positional encoding is evaluated inside the kernel; 1 = `raydirs` is SKYMLP.forward's
 * own argument x, dev f32 [n_rays,33] rows the caller already encoded (voxlib.positional_encoding; hidden_terms must be 3) */
int sdn_sky_mlp(const float *raydirs, const void *packed, const float *consts, float *sky_c, float *sky_partial, int32_t n_rays,
                int32_t n_workgroups, float *sky_avg, uint32_t *counter, int32_t hidden_terms, int32_t encoded, sdn_stream_t stream);

/* test hook: C[32,32] = A[32,16] * B[16,32] through the MFMA operand layouts field.hip relies on */
int sdn_debug_mfma_probe(const float *A, const float *B, float *C, sdn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SDNATIVE_H */
