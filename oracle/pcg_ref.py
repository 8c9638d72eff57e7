"""CPU ORACLE of the scene ingestion (TEST INFRASTRUCTURE ONLY): a literal restatement of
PCGVoxelGenerator.next_world (imaginaire/model_utils/pcg_gen.py:83-174) on in-memory maps instead of the three
files it reads.  Pinned on the unmodified reference by tests/test_scene_cpu.py (needs_reference), which writes the
maps to a temporary world directory and runs the reference's own next_world beside this function."""
import random

import numpy as np
import torch
import torch.nn.functional as F


def next_world(height_map, semantic_map, tree_map, trees_models, sample_height=256, rng=random):
    """height_map float [S0,S1]; semantic_map, tree_map uint8 [S0,S1]; trees_models list of int32 tensors.
    Returns dict(voxel_t int32 [Hv,S0,S1], heightmap int64, current_height_map, current_semantic_map, trans_mat)."""
    height_map = np.array(height_map, copy=True)
    semantic_map = np.asarray(semantic_map)
    tree_map = np.asarray(tree_map)
    height_map[height_map < 0] = 0                                                                   # :94
    height_map = ((height_map - height_map.min()) / (1 - height_map.min()) * (sample_height - 1)).astype(np.int16)  # :95
    total_size = height_map.shape                                                                    # :97
    org_semantic_map = torch.from_numpy(semantic_map.copy())                                         # :100
    org_semantic_map[torch.from_numpy(tree_map != 255)] = 10                                         # :101
    chunk_trees_map = tree_map
    biome_trees_dict = {'desert': [], 'savanna': [5], 'twoodland': [1, 7], 'tundra': [], 'seasonal forest': [1, 2],
                        'rainforest': [1, 2, 3], 'temp forest': [4], 'temp rainforest': [0, 3], 'boreal': [5, 6, 7],
                        'water': []}                                                                 # :104-115
    biome2mclabels = torch.tensor([28, 9, 8, 1, 9, 8, 9, 8, 30, 26], dtype=torch.int32)              # :116
    biome_names = list(biome_trees_dict.keys())
    chunk_grid_x, chunk_grid_y = torch.meshgrid(torch.arange(total_size[0]), torch.arange(total_size[1]), indexing="ij")
    world_voxel_t = torch.zeros(sample_height, total_size[0], total_size[1]).to(torch.int32)        # :119
    chunk_height_map = torch.from_numpy(height_map.astype(int))[None, ...]                           # :121
    chunk_semantic_map = torch.from_numpy(semantic_map)
    chunk_semantic_map = biome2mclabels[chunk_semantic_map[None, ...].long().contiguous()]           # :123
    world_voxel_t = world_voxel_t.scatter_(0, chunk_height_map, chunk_semantic_map)                  # :124
    pad_num = 16
    for preproc_step in range(pad_num):                                                              # :126-127
        world_voxel_t = world_voxel_t.scatter(0, torch.clip(chunk_height_map + preproc_step + 1, 0, sample_height - 1),
                                              chunk_semantic_map)
    chunk_height_map = chunk_height_map + pad_num
    chunk_height_map = chunk_height_map[0]
    boundary_detect = 50
    for biome_id in range(biome2mclabels.shape[0]):                                                  # :134-159
        tree_pos_mask = torch.from_numpy(chunk_trees_map == biome_id)
        tree_pos_x = chunk_grid_x[tree_pos_mask]
        tree_pos_y = chunk_grid_y[tree_pos_mask]
        tree_pos_h = chunk_height_map[tree_pos_mask]
        selected_trees = biome_trees_dict[biome_names[biome_id]]
        if len(selected_trees) == 0:
            continue
        for idx in range(len(tree_pos_x)):
            if tree_pos_x[idx] < boundary_detect or tree_pos_x[idx] > total_size[0] - boundary_detect or \
                    tree_pos_y[idx] < boundary_detect or tree_pos_y[idx] > total_size[1] - boundary_detect or \
                    tree_pos_h[idx] > sample_height - boundary_detect:
                continue
            tree_id = rng.choice(selected_trees)
            m = trees_models[tree_id]
            h, x, y = int(tree_pos_h[idx]), int(tree_pos_x[idx]), int(tree_pos_y[idx])
            tmp = world_voxel_t[h: h + m.shape[0], x: x + m.shape[1], y: y + m.shape[2]]
            tmp_mask = (tmp == 0)
            world_voxel_t[h: h + m.shape[0], x: x + m.shape[1], y: y + m.shape[2]][tmp_mask] = \
                m[:tmp.shape[0], :tmp.shape[1], :tmp.shape[2]][tmp_mask]
    trans_mat = torch.eye(4)                                                                         # :160
    m_, h_ = torch.max((torch.flip(world_voxel_t, [0]) != 0).int(), dim=0, keepdim=False)           # :162
    heightmap = world_voxel_t.shape[0] - 1 - h_
    heightmap[m_ == 0] = 0
    gnd_level = heightmap.min()
    sky_level = heightmap.max() + 1
    current_height_map = (chunk_height_map / (sample_height - 1))[None, None, ...]                   # :167
    current_semantic_map = F.one_hot(org_semantic_map.to(torch.int64)).to(torch.float).permute(2, 0, 1)[None, ...]
    voxel_t = world_voxel_t[gnd_level:sky_level, :, :]
    trans_mat[0, 3] += gnd_level
    return dict(voxel_t=voxel_t, heightmap=heightmap, current_height_map=current_height_map,
                current_semantic_map=current_semantic_map, trans_mat=trans_mat)


def synthetic_world(S, seed, n_models=8):
    """Seeded BEV maps + tree models in the formats next_world reads (heights in [0,1), biome ids 0..9, tree map = biome
    id at tree positions and 255 elsewhere; models = int32 block-id boxes with holes)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, S), np.linspace(0, 1, S), indexing="ij")
    h = 0.35 + 0.2 * np.sin(5 * xx + seed) * np.cos(4 * yy) + 0.1 * np.sin(17 * xx * yy) + 0.02 * rng.standard_normal((S, S))
    height = np.clip(h, -0.05, 0.95).astype(np.float32)
    sem = (np.floor((np.sin(3 * xx + 1) + np.cos(2 * yy + seed) + 2) * 2.49) % 10).astype(np.uint8)
    tree = np.full((S, S), 255, np.uint8)
    pos = rng.random((S, S)) < 0.004
    tree[pos] = sem[pos]
    models = []
    for k in range(n_models):
        d = (int(rng.integers(5, 12)), int(rng.integers(3, 8)), int(rng.integers(3, 8)))
        m = np.zeros(d, np.int32)
        m[:, d[1] // 2, d[2] // 2] = 34 + (k % 6)                               # trunk (log ids 34-39)
        crown = rng.random((d[0] - d[0] // 2, d[1], d[2])) < 0.6
        m[d[0] // 2:][crown] = 58 + (k % 6)                                     # leaves (ids 58-63)
        models.append(torch.from_numpy(m))
    return height, sem, tree, models
