"""Scene ingestion and the compact scene volume.

The reference keeps a world as an int32 volume of Minecraft block ids (`PCGVoxelGenerator.voxel_t`,
imaginaire/model_utils/pcg_gen.py:119-174): 3.3 GB for a 2048^2 world, built on the host with torch scatter
calls and a Python loop over tree positions.  A world holds ~20 distinct ids, so here the volume is a uint8 array of
PALETTE INDICES plus an int32 palette: 4x less to keep in HBM, to walk (sdn_rvip_u8) and to broadcast to the other
ranks, while every consumer still sees the reference's block ids (voxel_id comes out of the ray marcher as int32 ids,
`Scene.voxel_t` expands the volume on demand for the drop-in op / the unmodified generator).

`ingest()` is PCGVoxelGenerator.next_world from its three BEV maps, with the volume written on the GPU
(csrc/scene.hip); `compact()` converts a volume that arrives in the reference's format.
"""
import ctypes
import random

import numpy as np
import torch

from . import capi

_l3 = ctypes.c_int64 * 3

# pcg_gen.py:104-116
BIOME_TREES = {"desert": [], "savanna": [5], "twoodland": [1, 7], "tundra": [], "seasonal forest": [1, 2],
               "rainforest": [1, 2, 3], "temp forest": [4], "temp rainforest": [0, 3], "boreal": [5, 6, 7], "water": []}
BIOME2MCLABELS = [28, 9, 8, 1, 9, 8, 9, 8, 30, 26]
PAD_NUM, BOUNDARY_DETECT, SAMPLE_HEIGHT = 16, 50, 256


class CompactScene:
    """Duck type of the reference's scene handle backed by the compact volume."""

    def __init__(self):
        self.voxel_u8 = None       # dev u8 [Hv, S0, S1] palette indices, 0 = empty
        self.palette = None        # dev int32 [256], palette[0] = 0
        self.heightmap = None      # CPU int64 [S0, S1]  (camctl.py:302-306)
        self.current_height_map = None
        self.current_semantic_map = None
        self.trans_mat = None
        self.sample_size = None
        self._voxel_t = None

    @property
    def voxel_t(self):
        """The reference's int32 block-id volume, expanded from the compact one on first use (and cached)."""
        if self._voxel_t is None:
            self._voxel_t = self.palette[self.voxel_u8.long()].to(torch.int32)
        return self._voxel_t

    def world2local(self, v, is_vec=False):   # pcg_gen.py:176-178 with trans_mat = I + gnd shift (as synth.Scene)
        v = torch.as_tensor(v, dtype=torch.float32).clone()
        if not is_vec:
            v[0] = v[0] - self.trans_mat[0, 3]
        return v


def make_palette(ids):
    """ids: iterable of the distinct non-zero block ids.  Returns (palette int32[256], id2idx uint8[max_id + 1])."""
    ids = sorted(int(i) for i in set(int(v) for v in ids) if i != 0)
    if len(ids) > 255:
        raise RuntimeError(f"{len(ids)} distinct block ids: the compact volume holds at most 255 (use the int32 volume)")
    palette = np.zeros(256, np.int32)
    palette[1:1 + len(ids)] = ids
    id2idx = np.zeros(max(ids + [0]) + 1, np.uint8)
    for k, v in enumerate(ids):
        id2idx[v] = k + 1
    return palette, id2idx


def compact(voxel_t):
    """int32 block-id volume on the GPU (any strides) -> (voxel_u8, palette) on the same device."""
    assert voxel_t.is_cuda and voxel_t.dtype == torch.int32 and voxel_t.dim() == 3
    dev = voxel_t.device
    palette, id2idx = make_palette(torch.unique(voxel_t).cpu().tolist())
    out = torch.empty(tuple(voxel_t.shape), dtype=torch.uint8, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    lut = torch.from_numpy(id2idx).to(dev)
    with torch.cuda.device(dev):
        capi.check(capi.lib().sdn_volume_compact(voxel_t.data_ptr(), _l3(*voxel_t.shape), _l3(*voxel_t.stride()), lut.data_ptr(),
                                                 int(lut.numel()), out.data_ptr(), bad.data_ptr(), capi.current_stream(dev)),
                   "sdn_volume_compact")
    if int(bad.item()):
        raise RuntimeError("sdn_volume_compact: the volume changed while it was being compacted")
    return out, torch.from_numpy(palette).to(dev)


def to_compact(scene):
    """A CompactScene holding the same world as `scene` (any object with the reference's scene-handle attributes and an
    int32 `voxel_t` on the GPU); the int32 volume is not kept."""
    if getattr(scene, "voxel_u8", None) is not None:
        return scene
    sc = CompactScene()
    sc.voxel_u8, sc.palette = compact(scene.voxel_t)
    sc.heightmap, sc.trans_mat, sc.sample_size = scene.heightmap, scene.trans_mat, scene.sample_size
    sc.current_height_map, sc.current_semantic_map = scene.current_height_map, scene.current_semantic_map
    return sc


def normalise_height_map(height_map):
    """pcg_gen.py:94-95."""
    hm = np.array(height_map, dtype=np.float64 if np.asarray(height_map).dtype == np.float64 else np.float32, copy=True)
    hm[hm < 0] = 0
    return ((hm - hm.min()) / (1 - hm.min()) * (SAMPLE_HEIGHT - 1)).astype(np.int16)


def place_trees(tree_map, top_height, tree_models, total_size, rng=random):
    """The reference's tree loop (pcg_gen.py:134-146) up to the paste: [(h, x, y, model)] in the reference's order, with
    `rng.choice` consulted exactly as the reference consults `random.choice` (so a seeded run picks the same models)."""
    out = []
    names = list(BIOME_TREES.keys())
    tree_map = np.asarray(tree_map)
    for biome_id in range(len(BIOME2MCLABELS)):
        selected = BIOME_TREES[names[biome_id]]
        if len(selected) == 0:
            continue
        xs, ys = np.nonzero(tree_map == biome_id)          # row-major order == boolean-mask indexing order
        for x, y in zip(xs.tolist(), ys.tolist()):
            h = int(top_height[x, y])
            if x < BOUNDARY_DETECT or x > total_size[0] - BOUNDARY_DETECT or y < BOUNDARY_DETECT or \
                    y > total_size[1] - BOUNDARY_DETECT or h > SAMPLE_HEIGHT - BOUNDARY_DETECT:
                continue
            out.append((h, x, y, rng.choice(selected)))
    return out


def _rounds(trees, model_dims):
    """Split the tree list into launches of mutually disjoint trees, preserving the reference's paste order between
    trees whose boxes overlap (an earlier tree keeps a shared cell)."""
    cell = 32
    grid = {}
    rounds = []
    level = []
    for k, (h, x, y, m) in enumerate(trees):
        d0, d1, d2 = model_dims[m]
        lv = 0
        keys = [(a, b) for a in range(x // cell, (x + d1 - 1) // cell + 1) for b in range(y // cell, (y + d2 - 1) // cell + 1)]
        for key in keys:
            for j in grid.get(key, ()):
                hj, xj, yj, mj = trees[j]
                e0, e1, e2 = model_dims[mj]
                if x < xj + e1 and xj < x + d1 and y < yj + e2 and yj < y + d2 and h < hj + e0 and hj < h + d0:
                    lv = max(lv, level[j] + 1)
        level.append(lv)
        for key in keys:
            grid.setdefault(key, []).append(k)
        while len(rounds) <= lv:
            rounds.append([])
        rounds[lv].append(k)
    return rounds


def ingest(height_map, semantic_map, tree_map, tree_models, device="cuda", rng=random):
    """PCGVoxelGenerator.next_world (pcg_gen.py:83-174) from the BEV maps: height_map float [S0,S1] (heightmap.npy),
    semantic_map / tree_map uint8 [S0,S1] (semanticmap.png, treemap.png), tree_models = pcg_asset['assets'] (int32 block-id
    arrays).  Returns a CompactScene on `device`."""
    dev = torch.device(device)
    lib = capi.lib()
    hm16 = normalise_height_map(height_map)                                    # :94-95
    S0, S1 = hm16.shape
    semantic_map = np.asarray(semantic_map, np.uint8)
    tree_map = np.asarray(tree_map, np.uint8)
    models = [np.asarray(t.cpu() if isinstance(t, torch.Tensor) else t, np.int32) for t in tree_models]
    palette, id2idx = make_palette(list(BIOME2MCLABELS) + [int(v) for m in models for v in np.unique(m)])
    col_idx = id2idx[np.asarray(BIOME2MCLABELS, np.int64)[semantic_map.astype(np.int64)]]    # :118
    vol = torch.zeros((SAMPLE_HEIGHT, S0, S1), dtype=torch.uint8, device=dev)
    st = capi.current_stream(dev)
    with torch.cuda.device(dev):
        h_dev = torch.from_numpy(hm16).to(dev)
        c_dev = torch.from_numpy(np.ascontiguousarray(col_idx)).to(dev)
        capi.check(lib.sdn_scene_columns(h_dev.data_ptr(), c_dev.data_ptr(), SAMPLE_HEIGHT, S0, S1, PAD_NUM, vol.data_ptr(), st),
                   "sdn_scene_columns")                                           # :119-128
        chunk_height = hm16.astype(np.int64) + PAD_NUM                            # :129-130
        trees = place_trees(tree_map, chunk_height, models, (S0, S1), rng)        # :134-146
        if trees:
            dims = [m.shape for m in models]
            offs = np.cumsum([0] + [int(np.prod(d)) for d in dims])[:-1].astype(np.int32)
            packed = torch.from_numpy(np.concatenate([id2idx[m.reshape(-1)] for m in models])).to(dev)
            offs_d = torch.from_numpy(offs).to(dev)
            dims_d = torch.from_numpy(np.asarray(dims, np.int32).reshape(-1)).to(dev)
            arr = np.asarray(trees, np.int32)
            for rnd in _rounds(trees, dims):                                      # :147-150 (paste where empty)
                t_dev = torch.from_numpy(np.ascontiguousarray(arr[rnd])).to(dev)
                capi.check(lib.sdn_scene_paste_trees(vol.data_ptr(), SAMPLE_HEIGHT, S0, S1, t_dev.data_ptr(), len(rnd),
                                                     packed.data_ptr(), offs_d.data_ptr(), dims_d.data_ptr(), st),
                           "sdn_scene_paste_trees")
        top = torch.empty((S0, S1), dtype=torch.int32, device=dev)
        capi.check(lib.sdn_scene_column_tops(vol.data_ptr(), SAMPLE_HEIGHT, S0, S1, top.data_ptr(), st),
                   "sdn_scene_column_tops")                                       # :162-164
    sc = CompactScene()
    heightmap = top.cpu().to(torch.int64)
    gnd, sky = int(heightmap.min()), int(heightmap.max()) + 1                     # :165-166
    org_sem = semantic_map.copy()
    org_sem[tree_map != 255] = 10                                                 # :100-101
    sc.current_height_map = (torch.from_numpy(chunk_height) / (SAMPLE_HEIGHT - 1))[None, None].to(torch.float32).to(dev)   # :167
    sc.current_semantic_map = torch.nn.functional.one_hot(torch.from_numpy(org_sem.astype(np.int64))).to(torch.float) \
        .permute(2, 0, 1)[None].to(dev)                                           # :168
    sc.heightmap = heightmap
    sc.voxel_u8 = vol[gnd:sky]                                                    # :173 (a view: dim 0 is the slow one)
    sc.palette = torch.from_numpy(palette).to(dev)
    sc.trans_mat = torch.eye(4)
    sc.trans_mat[0, 3] += gnd                                                     # :160, :174
    sc.sample_size = S0
    return sc
