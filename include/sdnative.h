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

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDN_ABI_VERSION 1

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
 * Results are bit-identical to the reference source evaluated without FMA
 * contraction (see oracle/sdn_oracle.c).
 */
int sdn_rvip(const int32_t *vox, const int64_t *dims, const int64_t *strides, const float *cam_ori,
             const float *cam_dir, const float *cam_up, float cam_f, const float *cam_c, const int *img_dims,
             int max_samples, int32_t *out_voxel_id, float *out_depth, float *out_raydirs, sdn_stream_t stream);

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
/*   grad [L,B,C]; grad_embeddings [sO,C] pre-zeroed by the caller (accumulated with atomics);
 *   grad_inputs [B,D] written when calc_grad_inputs.  F32 only in this release.          */
int sdn_grid_encode_bwd(const void *grad, const float *inputs, const void *embeddings, int emb_dtype,
                        const int32_t *offsets, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                        uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void *dy_dx,
                        void *grad_inputs, uint32_t gridtype, int align_corners, sdn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SDNATIVE_H */
