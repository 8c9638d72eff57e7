"""Deterministic evaluation camera trajectories (host side).

Mirrors the poses of the reference's EvalCameraController
(imaginaire/model_utils/gancraft/camctl.py:9-50 pattern 0, height helpers
:296-325) so that benchmarks render the reference's default orbit.  Poses are
known up front, which is what makes frame sharding across GPUs trivial.
The arithmetic is deliberately done with the same float32 torch CPU ops as the
reference so the poses agree to the bit.
"""
import numpy as np
import torch


def _get_height(heightmap, loc0, loc1, minheight):  # camctl.py:296-306
    loc0, loc1 = int(loc0), int(loc1)
    height = minheight
    for dx in range(-3, 4):
        for dy in range(-3, 4):
            y, z = loc0 + dx, loc1 + dy
            if y < 0 or y >= heightmap.shape[0] or z < 0 or z >= heightmap.shape[1]:
                height = max(height, minheight)
            else:
                height = max(height, heightmap[y, z] + 2)
    return height


def _filtfilt(h, decay):  # camctl.py:308-325: forward/backward leaky max
    out, prev = [], h[0]
    for v in h:
        prev = prev - decay
        if prev < v:
            prev = v
        out.append(prev)
    prev = h[-1]
    for i in range(len(h) - 1, -1, -1):
        prev = prev - decay
        if prev < h[i]:
            prev = h[i]
        out[i] = max(prev, out[i])
    return out


def eval_camera_poses(scene, maxstep=40, pattern=0, cam_ang=72, smooth_decay_multiplier=None):
    """List of (cam_ori f32[3], cam_dir f32[3], cam_up f32[3], cam_f float) in voxel-local coordinates.

    cam_f is the focal length for a unit-width image; the renderer multiplies it by (W-1)
    (scenedreamer.py:575).  smooth_decay_multiplier defaults to 150/maxstep (scenedreamer.py:565-567).
    """
    if pattern != 0:
        raise NotImplementedError("only the orbit pattern 0 is mirrored so far")
    if smooth_decay_multiplier is None:
        smooth_decay_multiplier = 150 / maxstep
    sy, sz = scene.voxel_t.size(1), scene.voxel_t.size(2)
    circle = torch.linspace(0, 2 * np.pi, steps=maxstep)
    size = min(sy, sz) / 2
    shift = size * 0.2
    size = size * 0.8
    heights = []
    for i in range(maxstep):
        far = torch.tensor([70, torch.sin(circle[i]) * size + sy / 2 + shift,
                            torch.cos(circle[i]) * size + sz / 2 + shift])
        heights.append(_get_height(scene.heightmap, far[1], far[2], far[0]))
    heights = _filtfilt(heights, decay=0.2 * smooth_decay_multiplier)
    poses = []
    for i in range(maxstep):
        far = torch.tensor([70, torch.sin(circle[i]) * size + sy / 2 + shift,
                            torch.cos(circle[i]) * size + sz / 2 + shift])
        far[0] = heights[i]
        near = torch.tensor([60, torch.sin(circle[i] + 0.5 * np.pi) * size * 0.5 + sy / 2 + shift,
                             torch.cos(circle[i] + 0.5 * np.pi) * size * 0.5 + sz / 2 + shift])
        cam_ori = scene.world2local(far)
        cam_dir = scene.world2local(near - far, is_vec=True)
        cam_up = scene.world2local(torch.tensor([1, 0, 0], dtype=torch.float32), is_vec=True)
        cam_f = 0.5 / np.tan(np.deg2rad(cam_ang / 2))
        poses.append((cam_ori, cam_dir, cam_up, cam_f))
    return poses


def frame_intrinsics(cam_f, resolution_hw, pad):
    """Per-frame intrinsics of inference_givenstyle (scenedreamer.py:553-554, :575-576)."""
    H, W = resolution_hw
    cam_res = [H + pad, W + pad]
    f = cam_f * (W - 1)  # so that the view does not depend on the padding
    c = [(cam_res[0] - 1) / 2, (cam_res[1] - 1) / 2]
    return f, c, cam_res


def tile_grid(cam_res, pad, tile_size=128):
    """Tile rectangles (h0, h1, w0, w1) of the reference's render loop (scenedreamer.py:600-612)."""
    nh = (cam_res[0] - pad + tile_size - 1) // tile_size
    nw = (cam_res[1] - pad + tile_size - 1) // tile_size
    tiles = []
    for ih in range(nh):
        h0 = ih * tile_size
        h1 = min(ih * tile_size + tile_size + pad, cam_res[0])
        for iw in range(nw):
            w0 = iw * tile_size
            w1 = min(iw * tile_size + tile_size + pad, cam_res[1])
            tiles.append((h0, h1, w0, w1))
    return tiles, nh, nw
