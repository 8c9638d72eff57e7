"""Deterministic evaluation camera trajectories (host side).

Mirrors the poses of the reference's EvalCameraController
(imaginaire/model_utils/gancraft/camctl.py:9-293, all ten patterns; height helpers
:296-325) so that benchmarks render the reference's trajectories.  Poses are
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


def _pose(scene, ori_world, target_world, cam_f):
    """(cam_ori, cam_dir, cam_up, cam_f) in voxel-local coordinates; camera up is the height axis (camctl.py:45-47)."""
    return (scene.world2local(ori_world), scene.world2local(target_world - ori_world, is_vec=True),
            scene.world2local(torch.tensor([1, 0, 0], dtype=torch.float32), is_vec=True), cam_f)


def _ring(angle, factors, cy, cz, height):
    """Point on a horizontal circle: [height, sin(a)*f0*f1*.. + cy, cos(a)*f0*f1*.. + cz].  The factors are applied
    left to right with float32 torch ops so that every rounding matches the reference's expression
    `torch.sin(a)*size*k*move[i] + centre` bit for bit."""
    sn, cs = torch.sin(angle), torch.cos(angle)
    for f in factors:
        sn, cs = sn * f, cs * f
    for a, b in zip(cy, cz):          # the centre is added term by term as well (+ size/2, then + shift)
        sn, cs = sn + a, cs + b
    return torch.tensor([height, sn, cs])


# Orbit family (camera_mode 0-5 of configs/scenedreamer_inference.yaml:2-7; camctl.py:20-205): the camera ("far" point)
# circles the scene at a terrain-following, filtfilt-smoothed height and looks at a "near" point on a smaller circle.
#   far_h / near_h : nominal heights;  sign : orbit direction;  move : radius schedule (start, end) or None
#   near_phase (x pi), near_scale : angular offset and radius factor of the look-at point
#   zoom : focal zoom schedule or None;  outward : camera sits on the near circle and looks at the far one (mode 5)
_ORBITS = {
    0: dict(far_h=70, near_h=60, sign=1.0, move=None, near_phase=0.5, near_scale=0.5, zoom=None, outward=False),
    1: dict(far_h=90, near_h=60, sign=1.0, move=None, near_phase=-0.3, near_scale=0.3, zoom=(1.0, 0.25), outward=False),
    2: dict(far_h=90, near_h=60, sign=1.0, move=(1.0, 0.2), near_phase=0.5, near_scale=0.3, zoom=None, outward=False),
    3: dict(far_h=70, near_h=60, sign=-1.0, move=(0.75, 0.2), near_phase=-0.4, near_scale=0.9, zoom=None, outward=False),
    4: dict(far_h=90, near_h=60, sign=1.0, move=(1.0, 0.5), near_phase=0.5, near_scale=0.3, zoom=None, outward=False),
    5: dict(far_h=60, near_h=60, sign=1.0, move=(1.0, 0.5), near_phase=0.5, near_scale=0.3, zoom=None, outward=True),
}


def _ground_extent(scene):
    """(size(1), size(2)) of the scene volume (camctl.py:15) without expanding a compact scene's int32 view."""
    v = getattr(scene, "voxel_u8", None)
    if v is None:
        v = scene.voxel_t
    return int(v.shape[1]), int(v.shape[2])


def _orbit_poses(scene, maxstep, cam_ang, decay, spec):
    sy, sz = _ground_extent(scene)
    circle = torch.linspace(0, 2 * np.pi, steps=maxstep)
    size = min(sy, sz) / 2
    shift = size * 0.2
    size = size * 0.8
    cy, cz = (sy / 2, shift), (sz / 2, shift)
    move = torch.linspace(spec["move"][0], spec["move"][1], steps=maxstep) if spec["move"] else None
    zoom = torch.linspace(spec["zoom"][0], spec["zoom"][1], steps=maxstep) if spec["zoom"] else None

    def far_point(i):
        fac = [size, move[i]] if move is not None else [size]
        return _ring(spec["sign"] * circle[i], fac, cy, cz, spec["far_h"])

    def near_point(i):
        fac = [size, spec["near_scale"], move[i]] if move is not None else [size, spec["near_scale"]]
        return _ring(spec["sign"] * circle[i] + spec["near_phase"] * np.pi, fac, cy, cz, spec["near_h"])

    cam_point = near_point if spec["outward"] else far_point
    look_point = far_point if spec["outward"] else near_point
    heights = []
    for i in range(maxstep):
        c = cam_point(i)
        heights.append(_get_height(scene.heightmap, c[1], c[2], c[0]))
    heights = _filtfilt(heights, decay=decay)
    poses = []
    for i in range(maxstep):
        c = cam_point(i)
        c[0] = heights[i]
        if zoom is not None:
            f = 0.5 / np.tan(np.deg2rad(cam_ang / 2) * zoom[i])
        else:
            f = 0.5 / np.tan(np.deg2rad(cam_ang / 2))
        poses.append(_pose(scene, c, look_point(i), f))
    return poses


def eval_camera_poses(scene, maxstep=40, pattern=0, cam_ang=72, smooth_decay_multiplier=None):
    """List of (cam_ori f32[3], cam_dir f32[3], cam_up f32[3], cam_f) in voxel-local coordinates for the reference's
    ten evaluation trajectories (EvalCameraController, camctl.py:9-293).

    cam_f is the focal length for a unit-width image; the renderer multiplies it by (W-1) (scenedreamer.py:575).
    smooth_decay_multiplier defaults to 150/maxstep (scenedreamer.py:565-567).  Poses are known up front, which is
    what makes frame sharding across GPUs trivial."""
    if smooth_decay_multiplier is None:
        smooth_decay_multiplier = 150 / maxstep
    sy, sz = _ground_extent(scene)
    if pattern in _ORBITS:
        return _orbit_poses(scene, maxstep, cam_ang, 0.2 * smooth_decay_multiplier, _ORBITS[pattern])
    circle = torch.linspace(0, 2 * np.pi, steps=maxstep)
    poses = []
    if pattern == 6:      # rise while looking down (camctl.py:206-228); fixed 73 deg FOV with a zoom ramp
        size = min(sy, sz) / 2 * 0.8
        lift = torch.linspace(0.0, 200.0, steps=maxstep)
        zoom = torch.linspace(0.8, 1.6, steps=maxstep)
        for i in range(maxstep):
            far = _ring(circle[i] / 4, [size, 0.2], (sy / 2, 0), (sz / 2, 0), 80 + lift[i])
            far[0] = _get_height(scene.heightmap, far[1], far[2], far[0])
            near = _ring(circle[i] / 4 + 0.5 * np.pi, [size, 0.1], (sy / 2, 0), (sz / 2, 0), 65)
            poses.append(_pose(scene, far, near, 0.5 / np.tan(np.deg2rad(73 / 2) * zoom[i])))
    elif pattern == 7:    # far away, 45 deg down, 19.5 deg FOV (camctl.py:229-248)
        rad = torch.tensor([np.deg2rad(45).astype(np.float32)])
        size = 1536
        for i in range(maxstep):
            far = torch.tensor([61 + size, torch.sin(rad) * size + sy / 2, torch.cos(rad) * size + sz / 2])
            near = torch.tensor([61, sy / 2, sz / 2])
            poses.append(_pose(scene, far, near, 0.5 / np.tan(np.deg2rad(19.5 / 2))))
    elif pattern == 8:    # lateral pan for perpetual view generation (camctl.py:250-266)
        size = sy // 2
        for i in range(maxstep):
            off = size // maxstep * (i - maxstep // 4)
            far = torch.tensor([300, 0 * size + sy // 2, -1 * size + sz / 2 + off])
            near = torch.tensor([120, 0 * size * 0.5 + sy // 2, -1 * size * 0.5 + sz / 2 + off])
            poses.append(_pose(scene, far, near, 0.5 / np.tan(np.deg2rad(cam_ang / 2))))
    elif pattern == 9:    # sliding-window fly-over (camctl.py:268-286)
        size = sz // 2
        for i in range(maxstep):
            far = torch.tensor([140, sy // 2, -size // 4 + size * 8 // maxstep * i], dtype=torch.float32)
            near = torch.tensor([100, sy // 2, size * 8 // maxstep * i], dtype=torch.float32)
            poses.append(_pose(scene, far, near, 0.5 / np.tan(np.deg2rad(cam_ang / 2))))
    else:
        raise ValueError(f"unknown camera pattern {pattern}")
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
