"""Seeded synthetic inputs: scene volume, network weights, style code.

Neither the pretrained checkpoint nor the terrain generator's dependencies are
available, so benchmarks and parity tests run on synthetic
data of the reference's shapes:

  * scene: a 17-voxel-thick height-field shell with biome labels, water and a
    few box trees, laid out exactly like PCGVoxelGenerator.next_world leaves it
    (imaginaire/model_utils/pcg_gen.py:83-174): voxel_t int32 [Hv, S, S] cropped
    to [gnd:sky], heightmap, current_height_map, current_semantic_map, trans_mat;
  * weights: every tensor of the generator's inference modules under the
    reference's state-dict names and shapes, variance-preserving random init,
    hash-grid embeddings U(-0.5, 0.5), biases U(-0.1, 0.1) so that numerical
    errors are visible;
  * style: z ~ N(0, 1)[1, 128].

All randomness comes from a counter-based integer hash (splitmix64 finaliser) so
every platform regenerates bit-identical data from (seed, name): golden fixtures
only need to store seeds.  Only +, *, floor and integer ops are used for the
scene so no libm call can perturb it.
"""
import zlib

import numpy as np
import torch

# --------------------------------------------------------------------------- RNG

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(x):
    x = (x ^ (x >> np.uint64(30))) * _M1
    x = (x ^ (x >> np.uint64(27))) * _M2
    return x ^ (x >> np.uint64(31))


def _stream_key(seed, name):
    return np.uint64((int(seed) * 0x100000001B3 + zlib.crc32(name.encode())) & 0xFFFFFFFFFFFFFFFF)


def hash_u01(seed, name, n, offset=0):
    """n float64 uniforms in [0,1), element i depends only on (seed, name, offset+i)."""
    with np.errstate(over="ignore"):
        key = _mix(_stream_key(seed, name) + _GOLD)
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        bits = _mix(idx * _GOLD + key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, name, shape, lo, hi, chunk=1 << 23):
    n = int(np.prod(shape))
    out = np.empty(n, np.float32)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        out[s:s + m] = (lo + (hi - lo) * hash_u01(seed, name, m, s)).astype(np.float32)
    return out.reshape(shape)


def normal(seed, name, shape, std=1.0):
    """Irwin-Hall(12) normal approximation: only adds of uniforms (bit-reproducible, no libm)."""
    n = int(np.prod(shape))
    acc = np.zeros(n, np.float64)
    for k in range(12):
        acc += hash_u01(seed, f"{name}#{k}", n)
    return ((acc - 6.0) * std).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------- scene

BIOME_TO_MC = np.asarray([28, 9, 8, 1, 9, 8, 9, 8, 30, 26], np.int32)  # pcg_gen.py:116
WATER_BIOME = 9
TREE_SEMANTIC = 10        # pcg_gen.py:103
TRUNK_ID, LEAF_ID = 34, 58  # minecraft ids whose reduced label is "tree" 
SAMPLE_HEIGHT = 256       # pcg_gen.py:79
SHELL_PAD = 16            # pcg_gen.py:124


def _value_noise(seed, name, S, cell):
    """Smooth value noise in [0,1) on an S x S grid, lattice spacing `cell` (float64, exact ops only)."""
    n = S // cell + 2
    lat = hash_u01(seed, name, n * n).reshape(n, n)
    c = np.arange(S, dtype=np.float64) / cell
    i = np.floor(c).astype(np.int64)
    f = c - i
    w = f * f * (3.0 - 2.0 * f)
    a = lat[i][:, i]
    b = lat[i][:, i + 1]
    c_ = lat[i + 1][:, i]
    d = lat[i + 1][:, i + 1]
    wy, wx = w[:, None], w[None, :]
    return (a * (1 - wx) + b * wx) * (1 - wy) + (c_ * (1 - wx) + d * wx) * wy


class Scene:
    """Duck type of the reference's `Generator.voxel` handle."""

    def __init__(self):
        self.voxel_t = None
        self.heightmap = None
        self.current_height_map = None
        self.current_semantic_map = None
        self.trans_mat = None
        self.sample_size = None

    def world2local(self, v, is_vec=False):  # pcg_gen.py:176-178 with trans_mat = I + gnd shift
        v = torch.as_tensor(v, dtype=torch.float32).clone()
        if not is_vec:
            v[0] = v[0] - self.trans_mat[0, 3]
        return v


def make_scene(S=2048, seed=3407, device="cpu", n_trees=None):
    """Build the synthetic scene.  2-D maps on the host (exact float64), the volume on `device`."""
    base = max(S // 8, 8)
    h = np.zeros((S, S), np.float64)
    amp, tot = 1.0, 0.0
    for o in range(4):
        h += amp * _value_noise(seed, f"height{o}", S, max(base >> o, 2))
        tot += amp
        amp *= 0.5
    h /= tot  # [0,1)
    h = np.clip((h - 0.25) * 2.0, 0.0, 0.999)  # stretch so that terrain spans a good part of the height range
    sea = 0.18
    water = h < sea
    h = np.where(water, sea, h)
    top = np.clip((h * (SAMPLE_HEIGHT - 1)).astype(np.int64), 0, SAMPLE_HEIGHT - 1)  # pcg_gen.py:94-95

    biome = np.floor(_value_noise(seed, "biome", S, max(S // 6, 4)) * 9.0).astype(np.int64)
    biome = np.clip(biome, 0, 8)
    biome[water] = WATER_BIOME
    label2d = BIOME_TO_MC[biome]

    dev = torch.device(device)
    top_t = torch.from_numpy(top).to(dev)
    lab_t = torch.from_numpy(label2d).to(dev)
    world = torch.zeros((SAMPLE_HEIGHT, S, S), dtype=torch.int32, device=dev)
    # shell: voxels top .. top+16 (clipped) carry the column's label (pcg_gen.py:122-127)
    for hh0 in range(0, SAMPLE_HEIGHT, 32):
        hh = torch.arange(hh0, min(hh0 + 32, SAMPLE_HEIGHT), device=dev).view(-1, 1, 1)
        hi = torch.clamp(top_t + SHELL_PAD, max=SAMPLE_HEIGHT - 1)
        m = (hh >= top_t) & (hh <= hi)
        world[hh0:hh0 + hh.shape[0]] = torch.where(m, lab_t, torch.zeros_like(lab_t)).to(torch.int32)
    surf = np.clip(top + SHELL_PAD, 0, SAMPLE_HEIGHT - 1)

    # trees: trunk column + leaf box, only into empty voxels (pcg_gen.py:150-153)
    if n_trees is None:
        n_trees = max(4, (S * S) // 4096)
    u = hash_u01(seed, "trees", 2 * n_trees)
    margin = min(50, S // 8)
    ty = (margin + u[:n_trees] * (S - 2 * margin - 6)).astype(np.int64)
    tz = (margin + u[n_trees:] * (S - 2 * margin - 6)).astype(np.int64)
    semantic = biome.copy()
    placed = 0
    for y, z in zip(ty.tolist(), tz.tolist()):
        if water[y, z] or surf[y, z] > SAMPLE_HEIGHT - 16:
            continue
        b = int(surf[y, z]) + 1
        blk = torch.zeros((9, 5, 5), dtype=torch.int32, device=dev)
        blk[0:6, 2, 2] = TRUNK_ID
        blk[5:9, :, :] = LEAF_ID
        blk[5:6, 2, 2] = TRUNK_ID
        tgt = world[b:b + 9, y:y + 5, z:z + 5]
        tgt[:] = torch.where(tgt == 0, blk, tgt)
        semantic[y, z] = TREE_SEMANTIC
        placed += 1
    if placed == 0:  # the one-hot below must have 11 channels (layers.py:28 expects 11)
        y = z = S // 2
        semantic[y, z] = TREE_SEMANTIC

    occ = world != 0
    any_col = occ.any(dim=0)
    idx = torch.arange(SAMPLE_HEIGHT, device=dev).view(-1, 1, 1)
    heightmap = torch.where(occ, idx, torch.zeros_like(idx)).amax(dim=0)
    heightmap = torch.where(any_col, heightmap, torch.zeros_like(heightmap))  # pcg_gen.py:162-164
    gnd = int(heightmap.min().item())
    sky = int(heightmap.max().item()) + 1

    sc = Scene()
    sc.sample_size = S
    sc.voxel_t = world[gnd:sky].contiguous()
    sc.heightmap = heightmap.cpu()
    chm = torch.from_numpy(((top + SHELL_PAD) / (SAMPLE_HEIGHT - 1)).astype(np.float32))[None, None]
    sem = torch.from_numpy(semantic)
    csm = torch.nn.functional.one_hot(sem, 11).to(torch.float32).permute(2, 0, 1)[None]
    sc.current_height_map = chm.to(dev)
    sc.current_semantic_map = csm.contiguous().to(dev)
    sc.trans_mat = torch.eye(4)
    sc.trans_mat[0, 3] += gnd
    return sc


# --------------------------------------------------------------------------- weights

HIDDEN = 256
STYLE_DIMS = 128
INTERM_STYLE = 256
FEAT_DIM = 64
NUM_LABELS = 12
SKY_IN = 33  # 3 * (2*5) + 3  (pe_lvl_raydir_sky=5, incl_orig)
GRID_CFG = dict(input_dim=5, num_levels=16, level_dim=8, base_resolution=16, log2_hashmap_size=19,
                desired_resolution=2048)  # scenedreamer.py:51


def _lin(seed, name, out_f, in_f, bias=True, fan_in=None, shape=None):
    fan_in = fan_in or in_f
    std = float(np.sqrt(2.0 / (1.0 + 0.2 ** 2) / fan_in))  # variance preserving under LeakyReLU(0.2)
    w = normal(seed, name + ".weight", shape or (out_f, in_f), std)
    d = {name + ".weight": w}
    if bias:
        d[name + ".bias"] = uniform(seed, name + ".bias", (out_f,), -0.1, 0.1)
    return d


def _modlin(seed, name):
    d = {
        name + ".weight": normal(seed, name + ".weight", (HIDDEN, HIDDEN), 1.0 / np.sqrt(HIDDEN)),
        name + ".weight_alpha": normal(seed, name + ".weight_alpha", (HIDDEN, INTERM_STYLE), 1.0 / np.sqrt(INTERM_STYLE)),
        name + ".bias_alpha": 1.0 + uniform(seed, name + ".bias_alpha", (HIDDEN,), -0.1, 0.1),
        name + ".weight_beta": normal(seed, name + ".weight_beta", (HIDDEN, INTERM_STYLE), 1.0 / np.sqrt(INTERM_STYLE)),
        name + ".bias_beta": uniform(seed, name + ".bias_beta", (HIDDEN,), -0.1, 0.1),
    }
    return d


def make_weights(seed=0, grid_log2_hashmap=19, with_embeddings=True):
    """dict name -> float32 ndarray, names/shapes as in the reference generator's state dict."""
    from .gridencoder import level_offsets
    w = {}
    # hash grid (gridencoder/grid.py:113-129)
    cfg = dict(GRID_CFG, log2_hashmap_size=grid_log2_hashmap)
    pls = np.exp2(np.log2(cfg["desired_resolution"] / cfg["base_resolution"]) / (cfg["num_levels"] - 1))
    offs = level_offsets(cfg["input_dim"], cfg["num_levels"], pls, cfg["base_resolution"], cfg["log2_hashmap_size"],
                         False)
    w["hash_encoder.offsets"] = offs
    if with_embeddings:
        w["hash_encoder.embeddings"] = uniform(seed, "hash_encoder.embeddings", (int(offs[-1]), cfg["level_dim"]),
                                               -0.5, 0.5)
    # render MLP (layers.py:60-90)
    w.update(_lin(seed, "render_net.fc_m_a", HIDDEN, NUM_LABELS, bias=False, fan_in=1))
    w["render_net.fc_m_a.weight"] *= np.float32(0.5)
    w.update(_lin(seed, "render_net.fc_1", HIDDEN, 128))
    for i in (2, 3, 4, 5, 6):
        w.update(_modlin(seed, f"render_net.fc_{i}"))
    w.update(_lin(seed, "render_net.fc_sigma", 1, HIDDEN))
    # a trained field has densities of order 10-100 (free energy = relu(sigma) * 0.25 * ~0.1 voxel);
    # without this scale every ray would be almost transparent and the field invisible in the output
    w["render_net.fc_sigma.weight"] *= np.float32(150.0)
    w["render_net.fc_sigma.bias"] *= np.float32(150.0)
    w["render_net.fc_sigma.bias"] += np.float32(75.0)  # centres sigma around 0 for the default seeds (empirical)
    w.update(_lin(seed, "render_net.fc_out_c", FEAT_DIM, HIDDEN))
    # sky MLP (gancraft_base.py:132-148)
    w.update(_lin(seed, "sky_net.fc_z_a", HIDDEN, INTERM_STYLE, bias=False))
    w["sky_net.fc_z_a.weight"] *= np.float32(0.25)
    w.update(_lin(seed, "sky_net.fc1", HIDDEN, SKY_IN))
    for i in (2, 3, 4, 5):
        w.update(_lin(seed, f"sky_net.fc{i}", HIDDEN, HIDDEN))
    w.update(_lin(seed, "sky_net.fc_out_c", FEAT_DIM, HIDDEN))
    w["sky_net.fc_out_c.weight"] *= np.float32(0.5)
    # style MLP (gancraft_base.py:94-111)
    w.update(_lin(seed, "style_net.fc_layers.0", HIDDEN, STYLE_DIMS))
    for i in range(1, 5):
        w.update(_lin(seed, f"style_net.fc_layers.{i}", HIDDEN, HIDDEN))
    w.update(_lin(seed, "style_net.fc_out", INTERM_STYLE, HIDDEN))
    w["style_net.fc_out.weight"] *= np.float32(4.0)  # style code of O(0.5) so the modulation matters
    # scene encoder (layers.py:25-39)
    w.update(_lin(seed, "world_encoder.sconv_head", 8, 11, fan_in=11 * 9, shape=(8, 11, 3, 3)))
    w.update(_lin(seed, "world_encoder.hconv_head", 8, 1, fan_in=9, shape=(8, 1, 3, 3)))
    c = 16
    for i in range(5):
        w.update(_lin(seed, f"world_encoder.conv_blocks.{i}.layers.0", c, c, bias=False, fan_in=c * 9, shape=(c, c, 3, 3)))
        w.update(_lin(seed, f"world_encoder.conv_blocks.{i}.layers.2", 2 * c, c, bias=False, fan_in=c * 9,
                      shape=(2 * c, c, 3, 3)))
        c *= 2
    w.update(_lin(seed, "world_encoder.fc1", 16, c))
    w.update(_lin(seed, "world_encoder.fc2", 2, 16))
    # render CNN (gancraft_base.py:175-195)
    w.update(_lin(seed, "denoiser.fc_z_cond", 4 * HIDDEN, INTERM_STYLE))
    w["denoiser.fc_z_cond.weight"] *= np.float32(0.25)
    w.update(_lin(seed, "denoiser.conv1", HIDDEN, FEAT_DIM, shape=(HIDDEN, FEAT_DIM, 1, 1)))
    for n, bias in (("conv2a", True), ("conv2b", False), ("conv3a", True), ("conv3b", False)):
        w.update(_lin(seed, "denoiser." + n, HIDDEN, HIDDEN, bias=bias, fan_in=HIDDEN * 9, shape=(HIDDEN, HIDDEN, 3, 3)))
    w["denoiser.conv2b.weight"] *= np.float32(0.5)
    w["denoiser.conv3b.weight"] *= np.float32(0.5)
    w.update(_lin(seed, "denoiser.conv4a", HIDDEN, HIDDEN, shape=(HIDDEN, HIDDEN, 1, 1)))
    w.update(_lin(seed, "denoiser.conv4b", HIDDEN, HIDDEN, shape=(HIDDEN, HIDDEN, 1, 1)))
    w.update(_lin(seed, "denoiser.conv4", 3, HIDDEN, shape=(3, HIDDEN, 1, 1)))
    w["denoiser.conv4.weight"] *= np.float32(0.25)  # keep tanh out of saturation
    return w


def fog_weights(weights, gain=0.03, level=6.0):
    """A copy of `weights` whose density is a thin fog: sigma' = gain * sigma + level.  With the synthetic density head
    (sigma ~ N(-4, 21)) that is 6 +- 0.65: strictly positive at every sample, and small enough that a ray's transmittance
    after all its samples (free energy <= 10.5 * 3.0 * 0.25 * 24 / 26 = 7.3 -> T >= 7e-4) stays above the early-termination
    threshold.  The regime in which NOTHING is removed: no ray terminates, no sample has weight zero, every colour of every
    sample is composited (volum_rendering_relu, mc_utils.py:154-161) -- the floor of the frame rate and the worst case for
    accumulated colour error."""
    w = dict(weights)
    w["render_net.fc_sigma.weight"] = (np.asarray(weights["render_net.fc_sigma.weight"], np.float32) * np.float32(gain)).astype(np.float32)
    w["render_net.fc_sigma.bias"] = (np.asarray(weights["render_net.fc_sigma.bias"], np.float32) * np.float32(gain) + np.float32(level)).astype(np.float32)
    return w


def make_style(seed=8888):
    return normal(seed, "style_z", (1, STYLE_DIMS), 1.0)
