"""Multi-GPU layer: one process per GPU, frames of a camera trajectory sharded across ranks.

The path shards by independent units (frames are pure functions of broadcast-once state and a
host-computed pose), so there is NO data-path collective: RCCL (torch.distributed
backend "nccl" on ROCm) is only used once, before rendering, to broadcast the scene volume, the
weights and the style code from rank 0.  Large tensors are sent as scatter + all_gather so that the
root pushes a distinct 1/N slice over each of its xGMI links instead of N-1 full copies (xGMI is
point-to-point; a flat broadcast is root-egress bound).

Backends: "nccl" (= RCCL) with device tensors is the production path.  "gloo" is supported for CPU tensors (how the
logic is tested without GPUs) AND for device tensors: gloo has no scatter / gather / all_gather on device memory, so
with gloo every collective on a device tensor is staged through host memory (`_host_staged`).  That is how the GPU-side
paths (compact-volume broadcast into device memory, sharded frames, tile-parallel frame with the real Renderer) run
under world_size 2 on a ONE-GPU box (tests/test_dist_gpu.py); the choice depends only on (backend, device type), which is
the same on every rank, never on a per-call try/except.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_frames(frames, rank, world):
    """Round-robin frame assignment: frame f is rendered by rank f % world."""
    return list(frames)[rank::world]


def _is_init():
    return dist.is_available() and dist.is_initialized()


def _host_staged(t):
    """gloo + device tensor: the collective runs on a host copy (gloo implements scatter / gather / all_gather for CPU
    tensors only).  A function of (backend, device type): identical on every rank."""
    return dist.get_backend() == "gloo" and t.is_cuda


def _scatter_supported(t):
    """Whether the scatter + all_gather form is used for tensor `t`.  Decided IDENTICALLY on every rank from the backend
    name and the tensor's device type -- never per call with try/except around a collective: if one rank raised and fell
    back to broadcast while the others sat in scatter, the job would hang.  nccl: device tensors only (a CPU tensor would
    have no implementation at all -> plain broadcast is not available either, so it is refused up front); gloo: CPU
    tensors, device tensors through their host copies."""
    b = dist.get_backend()
    if b == "nccl":
        if not t.is_cuda:
            raise RuntimeError("the nccl (RCCL) backend moves device tensors only; got a CPU tensor")
        return True
    return b == "gloo"


def all_reduce(t, op=None, group=None):
    """dist.all_reduce that also works for device tensors under gloo (host-staged)."""
    op = dist.ReduceOp.SUM if op is None else op
    if _host_staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def gather(t, dst=0, group=None):
    """List of every rank's `t` on rank `dst` (None elsewhere); device tensors under gloo are host-staged and returned on
    `t`'s device."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    src = t.cpu() if _host_staged(t) else t
    outs = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
    dist.gather(src, outs, dst=dst, group=group)
    if rank != dst:
        return None
    return [o.to(t.device) for o in outs] if src is not t else outs


def broadcast_large(t, src=0, min_numel=1 << 20):
    """Broadcast tensor `t` (allocated with the right shape/dtype on every rank) from `src`."""
    world = dist.get_world_size()
    if world == 1:
        return t
    flat = t.reshape(-1)
    n = flat.numel()
    rank = dist.get_rank()
    if _host_staged(t):
        # gloo + device memory: the same protocol on a (pinned) host image of the tensor, then one H2D copy
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        if rank == src:
            h.copy_(t)
        broadcast_large(h, src, min_numel)
        if rank != src:
            t.copy_(h)
        return t
    if n < min_numel or not t.is_contiguous() or not _scatter_supported(t):
        dist.broadcast(t, src)
        return t
    # scatter + all_gather in `world` equal chunks; a tensor whose size is not a multiple of `world` is sent in chunks of
    # ceil(n / world) elements with the last one zero-padded (it used to fall back to the flat, root-egress-bound broadcast)
    chunk = -(-n // world)
    mine = torch.empty(chunk, dtype=t.dtype, device=t.device)
    parts = None
    if rank == src:
        parts = [flat[i * chunk:(i + 1) * chunk] for i in range(world)]
        if parts[-1].numel() < chunk:
            last = torch.zeros(chunk, dtype=t.dtype, device=t.device)
            last[:parts[-1].numel()] = parts[-1]
            parts[-1] = last
        parts = [p.contiguous() for p in parts]
    dist.scatter(mine, parts, src=src)
    outs = [torch.empty(chunk, dtype=t.dtype, device=t.device) for _ in range(world)]
    dist.all_gather(outs, mine)
    if rank != src:
        for i, o in enumerate(outs):
            m = min(chunk, n - i * chunk)
            if m > 0:
                flat[i * chunk:i * chunk + m].copy_(o[:m])
    return t


def checksum(t):
    """Order-independent 64-bit checksum of a tensor's bytes (integer sum of 32-bit words)."""
    b = t.contiguous().reshape(-1)
    if b.element_size() == 4:
        v = b.view(torch.int32)
    elif b.element_size() == 8:
        v = b.view(torch.int64)
    else:
        v = b.view(torch.uint8)
    return int(v.to(torch.int64).sum().item())


def broadcast_tensor_dict(d, dev, src=0, verify=True):
    """Broadcast a name -> tensor dict known only on `src`.  Returns the dict on every rank."""
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in d.items()]
    dist.broadcast_object_list(meta, src=src)
    out = {}
    for k, shape, dt in meta[0]:
        dtype = getattr(torch, dt)
        t = d[k].to(dev).contiguous() if rank == src else torch.empty(shape, dtype=dtype, device=dev)
        broadcast_large(t, src)
        out[k] = t
    if verify:
        sums = torch.tensor([checksum(v) for v in out.values()], dtype=torch.int64, device=dev)
        lo, hi = sums.clone(), sums.clone()
        all_reduce(lo, op=dist.ReduceOp.MIN)
        all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError("broadcast integrity check failed: checksums differ across ranks")
    return out


def broadcast_state(scene, weights, style, dev, src=0, compact=True, stats=None):
    """Rank `src` holds (scene, weights dict of ndarrays, style); everyone gets equivalent objects on `dev`.

    compact=True: the scene volume travels as uint8 palette indices + an int32[256] palette (scene.py: 0.8 GB instead of
    the 3.3 GB int32 volume of a 2048^2 world) and the receivers keep it in that form (Renderer walks it directly).
    A scene with more than 255 distinct block ids, or one on the CPU (gloo tests), is sent as int32.
    `stats` (optional dict) receives {"scene_volume_bytes": bytes of the volume as sent}."""
    from .synth import Scene
    rank = dist.get_rank()
    tensors = {}
    if rank == src:
        tensors.update({"w:" + k: torch.as_tensor(np.asarray(v)) for k, v in weights.items()})
        u8 = getattr(scene, "voxel_u8", None)
        if u8 is None and compact and scene.voxel_t.is_cuda:
            from . import scene as sc_mod
            try:
                u8, pal = sc_mod.compact(scene.voxel_t)
            except RuntimeError:            # more than 255 distinct ids: int32 it is
                u8 = None
        elif u8 is not None:
            pal = scene.palette
        if u8 is not None:
            tensors["s:voxel_u8"] = u8
            tensors["s:palette"] = pal
        else:
            tensors["s:voxel_t"] = scene.voxel_t
        tensors["s:heightmap"] = scene.heightmap.to(torch.int64)
        tensors["s:current_height_map"] = scene.current_height_map
        tensors["s:current_semantic_map"] = scene.current_semantic_map
        tensors["s:trans_mat"] = scene.trans_mat
        tensors["z:style"] = torch.as_tensor(np.asarray(style))
    got = broadcast_tensor_dict(tensors, dev, src)
    if "s:voxel_u8" in got:
        from .scene import CompactScene
        sc = CompactScene()
        sc.voxel_u8, sc.palette = got["s:voxel_u8"], got["s:palette"]
        vol = sc.voxel_u8
    else:
        sc = Scene()
        sc.voxel_t = got["s:voxel_t"]
        vol = sc.voxel_t
    if stats is not None:
        stats["scene_volume_bytes"] = vol.numel() * vol.element_size()
    sc.heightmap = got["s:heightmap"].cpu()
    sc.current_height_map = got["s:current_height_map"]
    sc.current_semantic_map = got["s:current_semantic_map"]
    sc.trans_mat = got["s:trans_mat"].cpu()
    sc.sample_size = int(vol.shape[1])
    w = {k[2:]: v for k, v in got.items() if k.startswith("w:")}
    return sc, w, got["z:style"].cpu().numpy()


# --------------------------------------------------------------------------------------------------------------
# Tile-parallel rendering of ONE frame (BASELINE.json config 5: 3840x2160 over 8 GPUs)
# --------------------------------------------------------------------------------------------------------------
def row_bands(height, world):
    """Contiguous output-row bands of equal HEIGHT, one per rank: [(row0, row1)] * world."""
    return [(height * k // world, height * (k + 1) // world) for k in range(world)]


MIN_BAND_ROWS = 8


def balanced_row_bands(costs, world, min_rows=MIN_BAND_ROWS):
    """Contiguous output-row bands of equal COST: costs[r] >= 0 is the relative work of output row r (Renderer.row_costs: the
    field kernel only visits rays that hit something, so sky rows are nearly free and equal-height bands leave the ground bands
    bounding the frame).  Boundary k is the first row where the running cost reaches k / world of the total; every band keeps at
    least min_rows rows.  Deterministic in `costs`: ranks that computed the same costs cut the same bands.
    Boundaries fall on multiples of 4 rows (when the height is one): a band's field launch then covers whole 8 x 4 pixel blocks
    (with the 4-px apron on both sides) and takes the blocked ray order (csrc/field.hip RayWindow) like a full frame."""
    costs = np.asarray(costs, dtype=np.float64)
    H = costs.shape[0]
    if world <= 1 or H < world * min_rows or not np.isfinite(costs).all() or costs.sum() <= 0:
        return row_bands(H, world)
    cum = np.cumsum(np.maximum(costs, 0.0))
    cuts = [0]
    for k in range(1, world):
        r = int(np.searchsorted(cum, cum[-1] * k / world, side="left")) + 1
        if H % 4 == 0 and min_rows % 4 == 0:
            r = int(round(r / 4.0)) * 4
        r = max(r, cuts[-1] + min_rows)
        r = min(r, H - (world - k) * min_rows)
        cuts.append(r)
    cuts.append(H)
    return [(cuts[k], cuts[k + 1]) for k in range(world)]


FEEDBACK_DAMPING = 0.7      # rebalance_scale: exponent on the measured / predicted ratio (1 = trust one frame completely)
FEEDBACK_CLAMP = (0.5, 2.0)  # ... and the range one update may move a band's rows by


def rebalance_scale(costs, scale, bands, band_ms, damping=FEEDBACK_DAMPING):
    """Measured feedback for the band cut: `costs * scale` predicted each band's share of the frame, `band_ms` is what the bands
    took.  Returns the new per-row multipliers: the rows of band k are multiplied by (measured share / predicted share) ** damping,
    clamped, and the result renormalised to mean 1.  A pure function of its arguments: every rank holds the same costs, scale,
    bands and (all-reduced) band_ms, so all ranks reach the same next cut without exchanging anything more.
    The static model (Renderer.row_costs: hits + MISS_COST x width) cannot know what the field does with a ray -- colour-branch
    skipping and early termination make the cost of a hit content-dependent -- and at 8 bands of a 4K frame it was off by 16 %
    (profiles/r05_dist_config5_8ranks_1gpu.json); consecutive frames of a trajectory look alike, so one measured frame corrects
    the next (tools/band_balance.py: imbalance per iteration on the real frame)."""
    costs = np.asarray(costs, np.float64)
    scale = np.ones_like(costs) if scale is None else np.asarray(scale, np.float64).copy()
    ms = np.asarray(band_ms, np.float64)
    pred = np.asarray([float((costs[a:b] * scale[a:b]).sum()) for a, b in bands])
    if not (np.isfinite(ms).all() and ms.sum() > 0 and pred.sum() > 0 and len(ms) == len(bands)):
        return scale
    ratio = (ms / ms.sum()) / np.maximum(pred / pred.sum(), 1e-12)
    ratio = np.clip(ratio ** damping, *FEEDBACK_CLAMP)
    for (a, b), r in zip(bands, ratio):
        scale[a:b] *= r
    return scale / scale.mean()


def render_frame_tile_parallel(renderer, pose, resolution_hw, num_samples, mode="fused", group=None, balance=True, stats=None):
    """Every rank renders one row band of the frame; the only exchange step of the path is the frame-wide sky mean:
    all_reduce(sum) of 64+1 numbers.  Rank 0 receives the stitched image [1,3,H,W]; the other ranks return None.

    balance: bands of equal estimated work (balanced_row_bands on renderer.row_costs(pose, hw): a 1/16-resolution ray cast every
    rank performs for itself -- bit-identical everywhere, so no exchange) instead of equal height.  balance="feedback": the
    estimate is additionally multiplied by the per-row factors the previous timed frames of this resolution produced
    (rebalance_scale on the all-reduced band times: the same update on every rank); needs `stats` -- the per-band device
    times are what it learns from, one host synchronisation per frame, which the gather to rank 0 implies anyway.
    stats: optional dict; receives "bands", and -- measured with device events around this rank's work, exchanged in one extra
    all_reduce of `world` numbers -- "band_ms" (per rank) and "imbalance" = max / mean.

    `renderer` needs band_prepare(pose, hw, row0, row1, mode) -> {"sky_sum" f64[64], "sky_cnt" int, ...} and
    band_finish(handle, sky_avg[1,64], num_samples) -> image rows; scenedreamer_amd.renderer.Renderer provides both (and
    row_costs)."""
    world = dist.get_world_size(group) if _is_init() else 1
    rank = dist.get_rank(group) if _is_init() else 0
    H, W = resolution_hw
    if (world > 1 and mode == "fused" and hasattr(renderer, "calibrate_style") and getattr(renderer, "field_gate", None) is None and
            getattr(renderer, "cnn_terms3x3", None) is None):
        # bands of one frame must not mix precisions: the per-style gates are decided for the job before the first band (every
        # rank reaches this point for the same frame with the same, still undecided, state -- the style was set on all of them)
        # (the frame's own resolution -- calibrate_one reduces frames above CAL_MAX_PIXELS by itself and measures on a window)
        agree_precision(renderer, pose, (H, W), num_samples, group)
    costs = fb = None
    if balance and world > 1 and hasattr(renderer, "row_costs"):
        costs = np.asarray(renderer.row_costs(pose, resolution_hw), np.float64)
        if balance == "feedback":
            fb = renderer.__dict__.setdefault("_band_feedback", {})
            sc = fb.get((H, W, world))
            bands = balanced_row_bands(costs * sc if sc is not None else costs, world)
        else:
            bands = balanced_row_bands(costs, world)
    else:
        bands = row_bands(H, world)
    row0, row1 = bands[rank]
    timed = stats is not None and torch.cuda.is_available() and getattr(renderer, "dev", torch.device("cpu")).type == "cuda"
    if timed:
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
    hd = renderer.band_prepare(pose, resolution_hw, row0, row1, mode)
    red = torch.cat([hd["sky_sum"].to(torch.float64).reshape(64),
                     torch.tensor([float(hd["sky_cnt"])], dtype=torch.float64, device=hd["sky_sum"].device)])
    if timed:
        e1.record()
    if world > 1:
        all_reduce(red, op=dist.ReduceOp.SUM, group=group)
    sky_avg = (red[:64] / red[64]).to(torch.float32).reshape(1, 64)
    if timed:
        e2.record()
    img = renderer.band_finish(hd, sky_avg, num_samples)
    if stats is not None:
        stats["bands"] = bands
        if timed:
            e3 = torch.cuda.Event(enable_timing=True)
            e3.record()
            e3.synchronize()
            mine = e0.elapsed_time(e1) + e2.elapsed_time(e3)       # this rank's own work: prepare + finish (the wait in the all_reduce excluded)
            v = torch.zeros(world, dtype=torch.float64, device=img.device)
            v[rank] = mine
            if world > 1:
                all_reduce(v, op=dist.ReduceOp.SUM, group=group)
            ms = [float(x) for x in v.tolist()]
            stats["band_ms"] = ms
            stats["imbalance"] = max(ms) / (sum(ms) / len(ms)) if sum(ms) > 0 else None
            if fb is not None:
                fb[(H, W, world)] = rebalance_scale(costs, fb.get((H, W, world)), bands, ms)
    if world == 1:
        return img
    hmax = max(b[1] - b[0] for b in bands)
    pad = torch.zeros((1, 3, hmax, W), dtype=img.dtype, device=img.device)
    pad[:, :, :img.shape[2]] = img
    outs = gather(pad, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([o[:, :, :b[1] - b[0]] for o, b in zip(outs, bands)], dim=2)


def agree_precision(renderer, pose, resolution_hw, num_samples, group=None, more_poses=()):
    """The renderer's per-style precision gates (Renderer.calibrate_style: colour layers fp6 / 3-term, 3x3 convolutions 1-term /
    3-term, fused path / fp32 fallback -- measured end to end against the fp32 frame) evaluated ONCE FOR THE JOB: every rank
    calibrates on the same frame (`pose`; renders are bit-reproducible, so every rank measures the same errors), the
    measurements are reduced with MAX, and every rank adopts the decisions that follow from the reduced values
    (Renderer.adopt_precision) -- bands of one frame, or frames of one trajectory, never mix precisions.  Explicit settings
    stay as they are.  more_poses: further poses measured and MAX-combined, as the single-process trajectory loop does with
    the middle pose of its trajectory (Renderer.calibrate_style) -- the same policy, so a multi-rank job cannot adopt a cheaper
    rung than one process would for the same style and trajectory.  Returns {"cnn": cnn_calibration, "field": field_gate}."""
    renderer.cnn_calibration = None
    renderer.field_gate = None
    renderer.colour_terms_auto = None
    renderer.sky_terms_auto = None
    meas = renderer.calibrate_style(pose, resolution_hw, num_samples, more_poses=more_poses)["measurements"]
    if _is_init() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        slots = [(a, k) for a in ("field_err", "image_err", "sky_err", "cnn_diffs") for k in sorted(meas.get(a) or {}, key=str)]
        slots += [(k, None) for k in ("colour_diff", "cnn_diff") if k in meas]
        v = torch.tensor([meas[a][b] if b is not None else meas[a] for a, b in slots], dtype=torch.float64, device=renderer.dev)
        all_reduce(v, op=dist.ReduceOp.MAX, group=group)
        for (a, b), x in zip(slots, v.tolist()):
            if b is not None:
                meas[a][b] = float(x)
            else:
                meas[a] = float(x)
        renderer.adopt_precision(meas)
        renderer.field_gate["agreed_over_ranks"] = world
        if renderer.cnn_calibration is not None:
            renderer.cnn_calibration["agreed_over_ranks"] = world
    return {"cnn": renderer.cnn_calibration, "field": renderer.field_gate}


def agree_cnn_precision(renderer, pose, resolution_hw, num_samples, group=None):
    """agree_precision, returning the render CNN's record only (the entry point of rounds 2-3)."""
    return agree_precision(renderer, pose, resolution_hw, num_samples, group)["cnn"]
