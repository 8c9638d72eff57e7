// Scene ingestion on the GPU: PCGVoxelGenerator.next_world (imaginaire/model_utils/pcg_gen.py:83-174)
// builds the int32 block-id volume on the host with torch scatter calls and a Python loop over tree positions
// (seconds for a 2048^2 world); here the volume is written directly in its COMPACT form -- uint8 palette indices,
// 4x smaller than the reference's int32 ids, what sdn_rvip_u8 walks and what the ranks receive -- by three kernels:
//   columns_kernel   terrain shell: cells h .. min(h + pad, Hs - 1) of column (x, y) get the column's biome block
//                    (pcg_gen.py:121-128: scatter at h, then `pad` scatters at clip(h + step + 1, 0, Hs - 1));
//   trees_kernel     pastes tree models where the world is still empty (:134-159).  The reference pastes in tree order,
//                    so an earlier tree keeps a cell a later one also covers; the host orders overlapping trees into
//                    rounds (one launch per round), trees inside a round are disjoint;
//   heights_kernel   top non-empty cell of every column (:162-164) for the camera controller and the crop levels.
//   compact_kernel   int32 ids -> palette indices for volumes that arrive in the reference's format.
#include "sdn_common.h"

namespace {

__global__ __launch_bounds__(256) void columns_kernel(uint8_t *__restrict__ vol, const int16_t *__restrict__ height,
                                                      const uint8_t *__restrict__ column_idx, int Hs, int S0, int S1, int pad) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)S0 * S1;
    if (c >= n) return;
    const int h = height[c];
    const uint8_t v = column_idx[c];
    // cells h, h+1, ..., h+pad, each clipped to [0, Hs-1] (the clipped ones all land on the top cell)
    for (int k = 0; k <= pad; k++) {
        int y = h + k;
        y = y < 0 ? 0 : (y > Hs - 1 ? Hs - 1 : y);
        vol[(int64_t)y * n + c] = v;
    }
}

struct Tree {
    int32_t h, x, y, model;
};

__global__ __launch_bounds__(256) void trees_kernel(uint8_t *__restrict__ vol, const Tree *__restrict__ trees,
                                                    const uint8_t *__restrict__ models, const int32_t *__restrict__ model_off,
                                                    const int32_t *__restrict__ model_dims, int Hs, int S0, int S1) {
    const Tree t = trees[blockIdx.x];
    const int d0 = model_dims[3 * t.model], d1 = model_dims[3 * t.model + 1], d2 = model_dims[3 * t.model + 2];
    const uint8_t *m = models + model_off[t.model];
    const int64_t n = (int64_t)S0 * S1;
    for (int i = threadIdx.x; i < d0 * d1 * d2; i += 256) {
        const int a = i / (d1 * d2), r = i - a * (d1 * d2), b = r / d2, c = r - b * d2;
        const int y = t.h + a, x = t.x + b, z = t.y + c;
        if (y >= Hs || x >= S0 || z >= S1) continue;      // python slicing clips at the array end
        const uint8_t v = m[i];
        if (v == 0) continue;
        uint8_t *cell = vol + (int64_t)y * n + (int64_t)x * S1 + z;
        if (*cell == 0) *cell = v;                         // trees of one launch are disjoint: no race
    }
}

__global__ __launch_bounds__(256) void heights_kernel(const uint8_t *__restrict__ vol, int32_t *__restrict__ top, int Hs, int64_t n) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    int h = 0;                                             // whole column empty -> 0 (pcg_gen.py:164)
    for (int y = Hs - 1; y >= 0; y--)
        if (vol[(int64_t)y * n + c] != 0) { h = y; break; }
    top[c] = h;
}

__global__ __launch_bounds__(256) void compact_kernel(const int32_t *__restrict__ vox, int64_t s0, int64_t s1, int64_t s2, int d1, int d2,
                                                      const uint8_t *__restrict__ id2idx, int n_ids, uint8_t *__restrict__ out,
                                                      int64_t n, int32_t *__restrict__ bad) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t a = i / ((int64_t)d1 * d2), r = i - a * ((int64_t)d1 * d2), b = r / d2, c = r - b * d2;
        const int32_t id = vox[a * s0 + b * s1 + c * s2];
        if (id < 0 || id >= n_ids || (id != 0 && id2idx[id] == 0)) {
            *bad = 1;          // an id the palette does not hold: reported to the caller, never mapped silently
            out[i] = 0;
        } else {
            out[i] = id2idx[id];
        }
    }
}

}  // namespace

extern "C" {

int sdn_scene_columns(const int16_t *height_map, const uint8_t *column_idx, int sample_height, int S0, int S1, int pad_num,
                      uint8_t *volume, sdn_stream_t stream) {
    SDN_REQUIRE(height_map && column_idx && volume, "sdn_scene_columns: null pointer");
    SDN_REQUIRE(sample_height > 0 && S0 > 0 && S1 > 0 && pad_num >= 0, "sdn_scene_columns: bad extent");
    const int64_t n = (int64_t)S0 * S1;
    hipLaunchKernelGGL(columns_kernel, dim3((unsigned)sdn::div_up<int64_t>(n, 256)), dim3(256), 0, (hipStream_t)stream, volume,
                       height_map, column_idx, sample_height, S0, S1, pad_num);
    return sdn::check_launch("sdn_scene_columns");
}

int sdn_scene_paste_trees(uint8_t *volume, int sample_height, int S0, int S1, const int32_t *trees_hxym, int n_trees,
                          const uint8_t *models, const int32_t *model_offsets, const int32_t *model_dims, sdn_stream_t stream) {
    SDN_REQUIRE(volume && models && model_offsets && model_dims, "sdn_scene_paste_trees: null pointer");
    if (n_trees <= 0) return SDN_OK;
    SDN_REQUIRE(trees_hxym, "sdn_scene_paste_trees: null tree list");
    hipLaunchKernelGGL(trees_kernel, dim3((unsigned)n_trees), dim3(256), 0, (hipStream_t)stream, volume, (const Tree *)trees_hxym,
                       models, model_offsets, model_dims, sample_height, S0, S1);
    return sdn::check_launch("sdn_scene_paste_trees");
}

int sdn_scene_column_tops(const uint8_t *volume, int sample_height, int S0, int S1, int32_t *top, sdn_stream_t stream) {
    SDN_REQUIRE(volume && top && sample_height > 0 && S0 > 0 && S1 > 0, "sdn_scene_column_tops: bad argument");
    const int64_t n = (int64_t)S0 * S1;
    hipLaunchKernelGGL(heights_kernel, dim3((unsigned)sdn::div_up<int64_t>(n, 256)), dim3(256), 0, (hipStream_t)stream, volume, top,
                       sample_height, n);
    return sdn::check_launch("sdn_scene_column_tops");
}

int sdn_volume_compact(const int32_t *vox, const int64_t *dims, const int64_t *strides, const uint8_t *id2idx, int n_ids,
                       uint8_t *out, int32_t *bad_flag, sdn_stream_t stream) {
    SDN_REQUIRE(vox && dims && strides && id2idx && out && bad_flag && n_ids > 0, "sdn_volume_compact: bad argument");
    SDN_REQUIRE(dims[0] > 0 && dims[1] > 0 && dims[2] > 0 && dims[1] < (1ll << 31) && dims[2] < (1ll << 31), "sdn_volume_compact: bad dims");
    const int64_t n = dims[0] * dims[1] * dims[2];
    hipLaunchKernelGGL(compact_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, vox, strides[0], strides[1], strides[2],
                       (int)dims[1], (int)dims[2], id2idx, n_ids, out, n, bad_flag);
    return sdn::check_launch("sdn_volume_compact");
}

}  // extern "C"
