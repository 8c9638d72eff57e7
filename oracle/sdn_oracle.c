/*
 * sdn_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the three native ops on SceneDreamer's inference hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (scenedreamer_amd/) never does.
 *
 * PARITY STATUS: PINNED on the reference's own sources.  The reference ships
 * no tests / golden vectors for these ops and nvcc is absent, but its three
 * .cu files compile for the HOST through a small shim of the CUDA language
 * features they use (oracle/build_ref.py -> oracle/_ref/, g++, real torch
 * headers).  tests/test_ref_pin_cpu.py demands identical BITS between this
 * restatement and that build for ray-voxel intersection (orbit poses on three
 * scenes, edge cases, strided volumes), grid encode forward + dy_dx (five
 * D/C/gridtype/align_corners instances incl. SceneDreamer's) and positional
 * encoding forward/backward; the scatter-add backward agrees to rounding.  The
 * golden field vectors regenerate exactly with the unmodified reference
 * (Python layers + oracle/_ref) on the CPU.  Further checks: analytic
 * known-answer tests, an independent vectorised numpy formulation
 * (tests/test_oracle_cpu.py) and the reference's own pure-torch twin for the
 * positional encoding (positional_encoding.py:45-54).  What stays unpinned is
 * nvcc's code generation itself (FMA contraction: see the "fma" variant of
 * oracle/_ref and DESIGN.md section 2).
 *
 * Floating point: built with -ffp-contract=off so every expression rounds
 * exactly as written in the reference source (no FMA contraction).  The HIP
 * ray-voxel kernel is built the same way, which is what makes the
 * voxel_id / depth2 / raydirs comparison bit-exact.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* vector helpers: imaginaire/model_utils/gancraft/voxlib/voxlib_common.h:26-74 */

static void v_cross(float *r, const float *a, const float *b) { /* :27-31 */
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}

static void v_normalize3(float *r, const float *a) { /* :49-73 */
    float vec_len = 0.0f;
    for (int i = 0; i < 3; i++) vec_len += a[i] * a[i];
    vec_len = sqrtf(vec_len);
    for (int i = 0; i < 3; i++) r[i] = a[i] / vec_len;
}

/* Camera frame, host side of ray_voxel_intersection.cu:279-286.
 * frame[0..2]=fwd, [3..5]=side, [6..8]=up */
ORACLE_API void oracle_camera_frame(const float *cam_dir, const float *cam_up, float *frame) {
    float fwd[3], side[3], up[3];
    v_normalize3(fwd, cam_dir);
    v_cross(side, fwd, cam_up);
    v_normalize3(side, side);
    v_cross(up, side, fwd);
    v_normalize3(up, up);
    memcpy(frame, fwd, 12);
    memcpy(frame + 3, side, 12);
    memcpy(frame + 6, up, 12);
}

/* ------------------------------------------------------------------------- */
/* Ray-voxel intersection: ray_voxel_intersection.cu:52-235 (device loop) and
 * :253-325 (host wrapper).  One call == one kernel launch.
 *
 *   vox        int32 volume, element (x,y,z) at vox[x*strides[0]+y*strides[1]+z*strides[2]]
 *   out_id     int32 [H, W, M]
 *   out_depth  f32   [2, H, W, M]
 *   out_dirs   f32   [H, W, 3]
 *   out_steps  optional int32 [H, W]: DDA iterations per ray (instrumentation only)
 */
ORACLE_API void oracle_rvip(const int32_t *vox, const int64_t *dims, const int64_t *strides,
                            const float *cam_ori, const float *cam_dir, const float *cam_up,
                            float cam_f, const float *cam_c, const int *img_dims, int max_samples,
                            int32_t *out_id, float *out_depth, float *out_dirs, int32_t *out_steps) {
    float frame[9];
    oracle_camera_frame(cam_dir, cam_up, frame);
    const float *fwd = frame, *side = frame + 3, *up = frame + 6;
    const int H = img_dims[0], W = img_dims[1], M = max_samples;
    const int64_t plane = (int64_t)H * W * M;
    const int vd[3] = {(int)dims[0], (int)dims[1], (int)dims[2]};

#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t pix = 0; pix < (int64_t)H * W; pix++) {
        const int r = (int)(pix / W), c = (int)(pix % W);
        float rayori[3] = {cam_ori[0], cam_ori[1], cam_ori[2]};
        float raydir[3];
        /* :66-78 camera intrinsics (height flipped) */
        float ndc0 = cam_c[0] - (float)r;
        float ndc1 = (float)c - cam_c[1];
        for (int i = 0; i < 3; i++) raydir[i] = up[i] * ndc0 + side[i] * ndc1 + fwd[i] * cam_f;
        v_normalize3(raydir, raydir);
        out_dirs[pix * 3 + 0] = raydir[0];
        out_dirs[pix * 3 + 1] = raydir[1];
        out_dirs[pix * 3 + 2] = raydir[2];

        float axis_t[3];
        int axis_int[3];
        for (int i = 0; i < 3; i++) axis_int[i] = (int)floorf(rayori[i]); /* :90-92 */
        for (int i = 0; i < 3; i++) {                                      /* :95-106 */
            if (raydir[i] > 0)
                axis_t[i] = ((float)(axis_int[i] + 1) - rayori[i]) / raydir[i];
            else if (raydir[i] < 0)
                axis_t[i] = ((float)axis_int[i] - rayori[i]) / raydir[i];
            else
                axis_t[i] = HUGE_VALF;
        }

        int quit = 0;
        int32_t steps = 0;
        for (int cur_plane = 0; cur_plane < M; cur_plane++) { /* :110 */
            float t = nanf("0"), t2 = nanf("0");
            int32_t blk_id = 0;
            while (!quit) { /* :115 */
                float tnow;
                int a; /* :143,:160,:175 axis choice with <= tie-breaks */
                if (axis_t[0] <= axis_t[1] && axis_t[0] <= axis_t[2]) a = 0;
                else if (axis_t[1] <= axis_t[2]) a = 1;
                else a = 2;
                tnow = axis_t[a];
                steps++;
                if (raydir[a] > 0) { /* :146-152 */
                    axis_int[a] += 1;
                    if (axis_int[a] >= vd[a]) quit = 1;
                    axis_t[a] = ((float)(axis_int[a] + 1) - rayori[a]) / raydir[a];
                } else { /* :153-159 */
                    axis_int[a] -= 1;
                    if (axis_int[a] < 0) quit = 1;
                    axis_t[a] = ((float)axis_int[a] - rayori[a]) / raydir[a];
                }
                if (quit) break; /* :192-194 */
                if (axis_int[0] < 0 || axis_int[0] >= vd[0] || axis_int[1] < 0 || axis_int[1] >= vd[1] ||
                    axis_int[2] < 0 || axis_int[2] >= vd[2])
                    continue; /* :198-200 */
                blk_id = vox[axis_int[0] * strides[0] + axis_int[1] * strides[1] + axis_int[2] * strides[2]];
                if (blk_id == 0) continue; /* :204-206 */
                t = tnow;                  /* :209 */
                if (axis_t[0] <= axis_t[1] && axis_t[0] <= axis_t[2]) t2 = axis_t[0]; /* :222-228 */
                else if (axis_t[1] <= axis_t[2]) t2 = axis_t[1];
                else t2 = axis_t[2];
                break;
            }
            out_depth[pix * M + cur_plane] = t;          /* :231 */
            out_depth[plane + pix * M + cur_plane] = t2; /* :232 */
            out_id[pix * M + cur_plane] = blk_id;        /* :233 (blk_id stays 0 on quit) */
        }
        if (out_steps) out_steps[pix] = steps;
    }
}

/* ------------------------------------------------------------------------- */
/* Positional encoding forward: positional_encoding_kernel.cu:40-75.
 * in [pre, post] -> out [pre, stride, post], stride = 2*ndeg (+1).           */
ORACLE_API void oracle_posenc_fwd(const float *in, float *out, int64_t pre, int64_t post, int ndeg,
                                  int incl_orig) {
    const float PI_F = 3.141592654f; /* CUDART_PI_F */
    int stride = ndeg * 2 + (incl_orig ? 1 : 0);
#pragma omp parallel for
    for (int64_t e = 0; e < pre; e++) {
        for (int64_t f = 0; f < post; f++) {
            float data = in[e * post + f];
            for (int i = 0; i < ndeg; i++) {
                float rad = data * PI_F * exp2f((float)i); /* :63 */
                out[e * post * stride + (int64_t)(i * 2) * post + f] = sinf(rad);
                out[e * post * stride + (int64_t)(i * 2 + 1) * post + f] = cosf(rad);
            }
            if (incl_orig) out[e * post * stride + (int64_t)(stride - 1) * post + f] = data; /* :71 */
        }
    }
}

/* Positional encoding backward: positional_encoding_kernel.cu:77-118 */
ORACLE_API void oracle_posenc_bwd(const float *out_grad, const float *out, float *in_grad, int64_t pre,
                                  int64_t post, int ndeg, int incl_orig) {
    const float PI_F = 3.141592654f;
    int stride = ndeg * 2 + (incl_orig ? 1 : 0);
#pragma omp parallel for
    for (int64_t e = 0; e < pre; e++) {
        for (int64_t f = 0; f < post; f++) {
            float grad = 0.0f;
            const int64_t base = e * post * stride + f;
            for (int i = 0; i < ndeg; i++) {
                float g = out_grad[base + (int64_t)(i * 2) * post] * out[base + (int64_t)(i * 2 + 1) * post];
                g -= out_grad[base + (int64_t)(i * 2 + 1) * post] * out[base + (int64_t)(i * 2) * post];
                grad += g * PI_F * exp2f((float)i);
            }
            if (incl_orig) grad += out_grad[base + (int64_t)(stride - 1) * post];
            in_grad[e * post + f] = grad;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Hash-grid encoder: gridencoder/src/gridencoder.cu                          */

static const uint32_t kPrimes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                    2097192037u, 1434869437u, 2165219737u}; /* :42 */

static uint32_t fast_hash(const uint32_t *pos_grid, uint32_t D) { /* :35-51 */
    uint32_t result = 0;
    for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * kPrimes[i];
    return result;
}

static uint32_t get_grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t ch,
                               uint32_t hashmap_size, uint32_t resolution, const uint32_t *pos_grid) { /* :54-72 */
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash(pos_grid, D);
    return (index % hashmap_size) * C + ch;
}

/* Exposed for known-answer tests. */
ORACLE_API uint32_t oracle_fast_hash(const uint32_t *pos_grid, uint32_t D) { return fast_hash(pos_grid, D); }
ORACLE_API uint32_t oracle_grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners,
                                      uint32_t hashmap_size, uint32_t resolution, const uint32_t *pos_grid) {
    return get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
}
ORACLE_API void oracle_level_params(uint32_t level, float S, uint32_t H, float *scale, uint32_t *resolution) {
    *scale = exp2f(level * S) * H - 1.0f;          /* :126 */
    *resolution = (uint32_t)ceil(*scale) + 1;      /* :127 */
}

static float f16r(float x);   /* float -> nearest-even half value (defined with the f16 backward below) */

/* kernel_grid<scalar_t,D,C>: gridencoder.cu:75-224.
 * inputs [B,D] in [0,1]; grid [sO,C]; offsets [L+1]; outputs [L,B,C];
 * dy_dx [B,L,D,C] when calc_grad_inputs.
 * half != 0: scalar_t = at::Half.  `results` / `results_grad` are halves there (:143, :184): every `+= w * grid[..]`
 * rounds the float product to half (implicit Half(float)) and then the half + half sum (c10::Half operator+), and
 * `grid[r] - grid[l]` is a half subtraction; the grid / outputs are passed as floats holding half values.          */
static void grid_fwd_impl(const float *inputs, const float *grid_all, const int32_t *offsets,
                          float *outputs_all, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                          float S, uint32_t H, int calc_grad_inputs, float *dy_dx_all,
                          uint32_t gridtype, int align_corners, int half) {
    for (uint32_t level = 0; level < L; level++) {
        const float *grid = grid_all + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = exp2f(level * S) * H - 1.0f;
        const uint32_t resolution = (uint32_t)ceil(scale) + 1;
#pragma omp parallel for
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const float *in = inputs + (size_t)b * D;
            float *outputs = outputs_all + (size_t)level * B * C + (size_t)b * C;
            float *dy_dx = calc_grad_inputs ? dy_dx_all + (size_t)b * D * L * C + (size_t)level * D * C : 0;
            int flag_oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (in[d] < 0 || in[d] > 1) flag_oob = 1; /* :99-106 */
            if (flag_oob) {
                for (uint32_t ch = 0; ch < C; ch++) outputs[ch] = 0;
                if (dy_dx)
                    for (uint32_t i = 0; i < D * C; i++) dy_dx[i] = 0;
                continue;
            }
            float pos[8];
            uint32_t pos_grid[8];
            for (uint32_t d = 0; d < D; d++) { /* :133-138 */
                pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
            }
            float results[8] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) { /* :146-171 */
                float w = 1;
                uint32_t pgl[8];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                uint32_t index = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    if (half) results[ch] = f16r(results[ch] + f16r(w * grid[index + ch]));
                    else results[ch] += w * grid[index + ch];
                }
            }
            for (uint32_t ch = 0; ch < C; ch++) outputs[ch] = results[ch];
            if (dy_dx) { /* :181-223 */
                for (uint32_t gd = 0; gd < D; gd++) {
                    float rg[8] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale;
                        uint32_t pgl[8];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                        }
                        pgl[gd] = pos_grid[gd];
                        uint32_t il = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                        pgl[gd] = pos_grid[gd] + 1;
                        uint32_t ir = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            if (half) rg[ch] = f16r(rg[ch] + f16r(w * f16r(grid[ir + ch] - grid[il + ch])));
                            else rg[ch] += w * (grid[ir + ch] - grid[il + ch]);
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++) dy_dx[gd * C + ch] = rg[ch];
                }
            }
        }
    }
}

ORACLE_API void oracle_grid_encode_fwd(const float *inputs, const float *grid_all, const int32_t *offsets,
                                       float *outputs_all, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                       float S, uint32_t H, int calc_grad_inputs, float *dy_dx_all,
                                       uint32_t gridtype, int align_corners) {
    grid_fwd_impl(inputs, grid_all, offsets, outputs_all, B, D, C, L, S, H, calc_grad_inputs, dy_dx_all, gridtype, align_corners, 0);
}

ORACLE_API void oracle_grid_encode_fwd_f16(const float *inputs, const float *grid_all, const int32_t *offsets,
                                           float *outputs_all, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                           float S, uint32_t H, int calc_grad_inputs, float *dy_dx_all,
                                           uint32_t gridtype, int align_corners) {
    grid_fwd_impl(inputs, grid_all, offsets, outputs_all, B, D, C, L, S, H, calc_grad_inputs, dy_dx_all, gridtype, align_corners, 1);
}

/* kernel_grid_backward (:227-314) + kernel_input_backward (:317-343), float only.
 * grad [L,B,C]; grad_grid [sO,C] (pre-zeroed by the caller); grad_inputs [B,D].
 * The reference scatters with atomicAdd in nondeterministic order; the oracle
 * accumulates sequentially in double and rounds once, so comparisons against it
 * use a tolerance, not bit equality.                                          */
ORACLE_API void oracle_grid_encode_bwd(const float *grad_all, const float *inputs, const int32_t *offsets,
                                       float *grad_grid_all, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                       float S, uint32_t H, int calc_grad_inputs, const float *dy_dx,
                                       float *grad_inputs, uint32_t gridtype, int align_corners) {
    size_t total = (size_t)(uint32_t)offsets[L] * C;
    double *acc = (double *)calloc(total, sizeof(double));
    for (uint32_t level = 0; level < L; level++) {
        double *gg = acc + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = exp2f(level * S) * H - 1.0f;
        const uint32_t resolution = (uint32_t)ceil(scale) + 1;
        for (uint32_t b = 0; b < B; b++) {
            const float *in = inputs + (size_t)b * D;
            const float *grad = grad_all + (size_t)level * B * C + (size_t)b * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;
            float pos[8];
            uint32_t pos_grid[8];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
            }
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[8];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                uint32_t index = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) gg[index + ch] += (double)(w * grad[ch]);
            }
        }
    }
    for (size_t i = 0; i < total; i++) grad_grid_all[i] += (float)acc[i];
    free(acc);
    if (calc_grad_inputs) {
        for (uint32_t t = 0; t < B * D; t++) {
            uint32_t b = t / D, d = t - b * D;
            const float *dd = dy_dx + (size_t)b * L * D * C;
            float result = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++)
                    result += grad_all[(size_t)l * B * C + (size_t)b * C + ch] * dd[l * D * C + d * C + ch];
            grad_inputs[t] = result;
        }
    }
}

/* ---- f16 variant of the backward pass (gridencoder.cu:296-304, :317-343 with scalar_t = at::Half) -------------
 * gcc 11 has no _Float16 on x86: round-to-nearest-even to half precision by hand.                                */
static float f16r(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return x;                         /* inf / nan */
    if (a >= 0x477ff000u) {                                 /* rounds to >= 65520 -> inf */
        a = 0x7f800000u;
    } else if (a < 0x38800000u) {                           /* below the smallest normal half 2^-14: subnormal grid 2^-24 */
        float ax;
        memcpy(&ax, &a, 4);
        const float q = ax * 16777216.0f;                   /* / 2^-24, exact */
        const float r = nearbyintf(q);                      /* default rounding mode: to nearest even */
        ax = r * (1.0f / 16777216.0f);
        memcpy(&a, &ax, 4);
    } else {
        const uint32_t lsb = (a >> 13) & 1u;
        a += 0x00000fffu + lsb;
        a &= 0xffffe000u;
    }
    a |= sign;
    float r;
    memcpy(&r, &a, 4);
    return r;
}

ORACLE_API float oracle_round_half(float x) { return f16r(x); }

/* grad / dy_dx hold half values (as floats).  grad_grid_all (+=): every contribution rounded to half like the
 * reference's (__half)(w * grad), then summed in double -- the reference adds them with half-precision atomics in
 * nondeterministic order, so comparisons use a tolerance.  grad_inputs: the reference's sequential half accumulation
 * (scalar_t result; result += grad * dy_dx), reproduced exactly.                                                  */
ORACLE_API void oracle_grid_encode_bwd_f16(const float *grad_all, const float *inputs, const int32_t *offsets,
                                           float *grad_grid_all, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                           float S, uint32_t H, int calc_grad_inputs, const float *dy_dx,
                                           float *grad_inputs, uint32_t gridtype, int align_corners) {
    size_t total = (size_t)(uint32_t)offsets[L] * C;
    double *acc = (double *)calloc(total, sizeof(double));
    for (uint32_t level = 0; level < L; level++) {
        double *gg = acc + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = exp2f(level * S) * H - 1.0f;
        const uint32_t resolution = (uint32_t)ceil(scale) + 1;
        for (uint32_t b = 0; b < B; b++) {
            const float *in = inputs + (size_t)b * D;
            const float *grad = grad_all + (size_t)level * B * C + (size_t)b * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;
            float pos[8];
            uint32_t pos_grid[8];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
            }
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[8];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                uint32_t index = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) gg[index + ch] += (double)f16r(w * grad[ch]);
            }
        }
    }
    for (size_t i = 0; i < total; i++) grad_grid_all[i] += (float)acc[i];
    free(acc);
    if (calc_grad_inputs) {
        for (uint32_t t = 0; t < B * D; t++) {
            uint32_t b = t / D, d = t - b * D;
            const float *dd = dy_dx + (size_t)b * L * D * C;
            float result = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++)
                    result = f16r(result + f16r(grad_all[(size_t)l * B * C + (size_t)b * C + ch] * dd[l * D * C + d * C + ch]));
            grad_inputs[t] = result;
        }
    }
}

ORACLE_API void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
