// Sin/cos positional encoding (forward + backward) for gfx950.
//
// Contract: voxlib.positional_encoding / positional_encoding_backward of the
// reference (imaginaire/model_utils/gancraft/voxlib/positional_encoding_kernel.cu:40-75,
// :77-118).  in [pre, post] -> out [pre, 2*ndeg(+1), post] with
// out[e, 2i, f] = sin(x * pi_f * 2^i), out[e, 2i+1, f] = cos(...), out[e, last, f] = x.
//
// The op is pure streaming (4 B in, (2*ndeg+1)*4 B out per element).  One lane
// owns one (entry, feature) element and walks the degrees, so that for every
// degree the wave writes 64 consecutive floats of the output row when post is
// large, and packs 64/post entries per wave when post is tiny (the sky ray
// directions have post = 3: a CUDA-style 16-wide x-tile would leave 13 of 16
// lanes idle there).
#include "sdn_common.h"

namespace {

constexpr float kPiF = 3.141592654f;  // CUDART_PI_F

__global__ __launch_bounds__(256) void posenc_fwd_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                         int64_t n, int64_t post, int ndeg, int incl_orig) {
    const int stride = ndeg * 2 + (incl_orig ? 1 : 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / post, f = i - e * post;
        const float x = in[i];
        float *o = out + e * post * stride + f;
        for (int d = 0; d < ndeg; d++) {
            const float rad = x * kPiF * exp2f((float)d);
            float s, c;
            sincosf(rad, &s, &c);
            o[(int64_t)(2 * d) * post] = s;
            o[(int64_t)(2 * d + 1) * post] = c;
        }
        if (incl_orig) o[(int64_t)(stride - 1) * post] = x;
    }
}

__global__ __launch_bounds__(256) void posenc_bwd_kernel(const float *__restrict__ og, const float *__restrict__ out,
                                                         float *__restrict__ ig, int64_t n, int64_t post, int ndeg,
                                                         int incl_orig) {
    const int stride = ndeg * 2 + (incl_orig ? 1 : 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / post, f = i - e * post;
        const int64_t base = e * post * stride + f;
        float grad = 0.0f;
        for (int d = 0; d < ndeg; d++) {
            // d/dx sin = cos * k, d/dx cos = -sin * k, k = pi * 2^d
            float g = og[base + (int64_t)(2 * d) * post] * out[base + (int64_t)(2 * d + 1) * post];
            g -= og[base + (int64_t)(2 * d + 1) * post] * out[base + (int64_t)(2 * d) * post];
            grad += g * kPiF * exp2f((float)d);
        }
        if (incl_orig) grad += og[base + (int64_t)(stride - 1) * post];
        ig[i] = grad;
    }
}

inline int grid_for(int64_t n) {
    int64_t blocks = sdn::div_up<int64_t>(n, 256);
    const int64_t cap = 256 * 8;  // 256 CUs x 8 resident blocks, grid-stride beyond
    return (int)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

}  // namespace

extern "C" int sdn_posenc_fwd(const float *in, float *out, int64_t pre, int64_t post, int ndegrees, int incl_orig,
                              sdn_stream_t stream) {
    SDN_REQUIRE(pre >= 0 && post >= 0 && ndegrees >= 0, "sdn_posenc_fwd: negative size");
    const int64_t n = pre * post;
    if (n == 0) return SDN_OK;
    SDN_REQUIRE(in && out, "sdn_posenc_fwd: null pointer");
    hipLaunchKernelGGL(posenc_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n, post,
                       ndegrees, incl_orig);
    return sdn::check_launch("sdn_posenc_fwd");
}

extern "C" int sdn_posenc_bwd(const float *out_grad, const float *out, float *in_grad, int64_t pre, int64_t post,
                              int ndegrees, int incl_orig, sdn_stream_t stream) {
    SDN_REQUIRE(pre >= 0 && post >= 0 && ndegrees >= 0, "sdn_posenc_bwd: negative size");
    const int64_t n = pre * post;
    if (n == 0) return SDN_OK;
    SDN_REQUIRE(out_grad && out && in_grad, "sdn_posenc_bwd: null pointer");
    hipLaunchKernelGGL(posenc_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out_grad, out,
                       in_grad, n, post, ndegrees, incl_orig);
    return sdn::check_launch("sdn_posenc_bwd");
}
