"""Multi-rank logic on CPU with the gloo backend (world_size 2): frame sharding and the scene/weights/style
broadcast with its integrity check."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from scenedreamer_amd import dist as sdist
        from scenedreamer_amd import synth
        scene = weights = style = None
        if rank == 0:
            scene = synth.make_scene(64, 11)
            weights = synth.make_weights(0, grid_log2_hashmap=10)
            style = synth.make_style(8888)
        sc, w, st = sdist.broadcast_state(scene, weights, style, torch.device("cpu"), src=0)
        ref_scene = synth.make_scene(64, 11)
        ref_w = synth.make_weights(0, grid_log2_hashmap=10)
        ok = torch.equal(sc.voxel_t, ref_scene.voxel_t) and torch.equal(sc.heightmap, ref_scene.heightmap.to(torch.int64))
        ok = ok and all(np.array_equal(w[k].numpy(), np.asarray(ref_w[k])) for k in ref_w) and set(w) == set(ref_w)
        ok = ok and np.array_equal(st, synth.make_style(8888))
        # large-tensor path: scatter + all_gather
        t = torch.arange(1 << 21, dtype=torch.int32) if rank == 0 else torch.empty(1 << 21, dtype=torch.int32)
        sdist.broadcast_large(t, 0)
        ok = ok and bool((t == torch.arange(1 << 21, dtype=torch.int32)).all())
        # ... whose size is NOT a multiple of the world size: padded last chunk, still scatter + all_gather
        n_odd = (1 << 20) + 3
        t = torch.arange(n_odd, dtype=torch.float32).reshape(-1) if rank == 0 else torch.full((n_odd,), -1.0)
        sdist.broadcast_large(t, 0)
        ok = ok and bool((t == torch.arange(n_odd, dtype=torch.float32)).all())
        frames = sdist.shard_frames(range(11), rank, world)
        q.put((rank, ok, frames))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_frame_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert all(ok for _, ok, _ in res)
    frames = [f for _, _, fr in res for f in fr]
    assert sorted(frames) == list(range(11))                          # every frame rendered exactly once
    assert res[0][2] == [0, 2, 4, 6, 8, 10] and res[1][2] == [1, 3, 5, 7, 9]


def test_checksum_detects_corruption():
    from scenedreamer_amd import dist as sdist
    a = torch.arange(1000, dtype=torch.float32)
    b = a.clone()
    b[17] += 1
    assert sdist.checksum(a) != sdist.checksum(b) and sdist.checksum(a) == sdist.checksum(a.clone())


class _FakeRenderer:
    """Stands in for Renderer in the orchestration test: sky feature of a ray = f(global padded row), image row
    value = global row index + frame-wide sky mean, so any band/ownership/stitching mistake changes the result."""
    pad = 30

    def __init__(self, H, W):
        self.H, self.W = H, W

    def band_prepare(self, pose, hw, row0, row1, mode):
        H, W = hw
        Wp = W + self.pad
        rows = torch.arange(row0, row1 + self.pad, dtype=torch.float64)
        sky_c = (rows[:, None, None] * 0.01 + torch.arange(64, dtype=torch.float64)[None, None, :]).expand(-1, Wp, -1).reshape(-1, 64)
        own = (row1 - row0 + (self.pad if row1 == H else 0)) * Wp
        return dict(sky_sum=sky_c[:own].sum(0), sky_cnt=own, rows=(row0, row1))

    def band_finish(self, hd, sky_avg, ns):
        r0, r1 = hd["rows"]
        img = torch.arange(r0, r1, dtype=torch.float32)[None, None, :, None].expand(1, 3, -1, self.W).clone()
        return img + sky_avg.reshape(-1)[:3].to(torch.float32)[None, :, None, None]


def _tp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from scenedreamer_amd import dist as sdist
        H, W = 37, 20          # ragged: 37 rows over 2 ranks -> 18 + 19
        img = sdist.render_frame_tile_parallel(_FakeRenderer(H, W), None, (H, W), 4)
        q.put((rank, None if img is None else img.numpy()))
    finally:
        dist.destroy_process_group()


def test_tile_parallel_single_frame_world2():
    from scenedreamer_amd import dist as sdist
    H, W = 37, 20
    single = sdist.render_frame_tile_parallel(_FakeRenderer(H, W), None, (H, W), 4).numpy()   # no process group: world 1
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[1] is None and res[0].shape == (1, 3, H, W)
    np.testing.assert_allclose(res[0], single, rtol=0, atol=1e-6)
    assert sdist.row_bands(37, 2) == [(0, 18), (18, 37)] and sdist.row_bands(2160, 8)[-1] == (1890, 2160)


def test_balanced_row_bands():
    """Bands of equal estimated work: a frame whose upper third is sky (cheap rows) gets a tall first band; degenerate cost
    vectors fall back to equal heights; every band keeps MIN_BAND_ROWS rows; the cuts are a pure function of the costs."""
    from scenedreamer_amd import dist as sdist
    W = 990
    costs = np.r_[np.full(200, 0.2 * W), np.full(340, 0.2 * W + 0.9 * W)]
    bands = sdist.balanced_row_bands(costs, 8)
    assert bands[0][0] == 0 and bands[-1][1] == 540 and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
    work = [costs[a:b].sum() for a, b in bands]
    assert max(work) / (sum(work) / 8) < 1.03 and bands[0][1] - bands[0][0] > 150
    equal = [costs[a:b].sum() for a, b in sdist.row_bands(540, 8)]
    assert max(equal) / (sum(equal) / 8) > 1.3                       # what equal-height bands would cost
    assert sdist.balanced_row_bands(costs, 8) == bands
    assert sdist.balanced_row_bands(np.zeros(100), 4) == sdist.row_bands(100, 4)
    assert sdist.balanced_row_bands(np.ones(20), 4) == sdist.row_bands(20, 4)           # too few rows for 4 bands of 8
    tail = sdist.balanced_row_bands(np.r_[np.zeros(90), np.ones(10)], 4)
    assert all(b - a >= sdist.MIN_BAND_ROWS for a, b in tail) and tail[-1][1] == 100
    assert sdist.balanced_row_bands(np.ones(540), 1) == [(0, 540)]


def _agree_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from scenedreamer_amd import dist as sdist
        from scenedreamer_amd import renderer as rmod

        R = object.__new__(rmod.Renderer)          # adopt_precision is a pure function of the measurements
        R.dev = torch.device("cpu")
        R._fused_style = None

        def calibrate_style(pose, hw, ns, more_poses=()):
            # what two ranks could measure on their frames: rank 1 sees the rung "1113" just outside the gate, rank 0 inside
            d1113 = 4.5e-4 if rank == 0 else 5.5e-4
            meas = dict(field_err={6: 6e-5 + 1e-5 * rank, 3: 5e-5}, colour_diff=5e-5, sky_err={3: 4e-6, 6: 1e-4},
                        image_err={1: 7e-4, "1113": d1113 + 1e-5, "1133": 3.5e-4, 3: 5e-5}, cnn_diff=6e-4,
                        cnn_diffs={1: 6e-4, "1113": d1113, "1133": 3.4e-4}, explicit_colour=None, explicit_cnn=None, explicit_sky=None,
                        pixels=518400, rays=564300, samples_per_ray=24, frame="t")
            return R.adopt_precision(meas)
        R.calibrate_style = calibrate_style
        own = calibrate_style(None, None, None)
        own_rung = R.cnn_calibration["terms3x3"]
        got = sdist.agree_precision(R, None, (540, 960), 24)
        q.put((rank, own_rung, got["cnn"]["terms3x3"], got["cnn"]["max_abs_diff_vs_3term"]["1113"], got["field"]["max_abs_err_vs_fp32"],
               got["cnn"].get("agreed_over_ranks")))
    finally:
        dist.destroy_process_group()


def test_precision_ladder_is_agreed_over_ranks_world2():
    """dist.agree_precision with the CNN ladder: every rank measures every rung (mixed int / str keys), the measurements are reduced
    with MAX, and all ranks adopt the same rung -- the one the WORST rank's measurements allow."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ["1113", "1133"]                 # left alone, the ranks would have rendered with different precisions
    assert [r[2] for r in res] == ["1133", "1133"]                 # agreed: the rung the worst measurement allows
    assert all(abs(r[3] - 5.5e-4) < 1e-12 and abs(r[4] - 7e-5) < 1e-12 and r[5] == 2 for r in res)


def test_band_feedback_converges_on_a_wrong_cost_model():
    """dist.rebalance_scale: the static row-cost model is wrong by a content-dependent factor (here: the true cost of a 'ground'
    row is 2.2x the model's in the lower third, sky rows cost 3x the model's) -- bands cut on the model are 25 % out of balance at
    8 bands; feeding the measured band times back (what render_frame_tile_parallel(balance="feedback") does between consecutive
    frames) brings max / mean below 1.05 within three frames, on cuts that stay multiples of 4 rows, identically on every rank
    (a pure function of shared values)."""
    import numpy as np
    from scenedreamer_amd import dist as sdist
    H, world = 2160, 8
    rows = np.arange(H)
    model = np.where(rows < 700, 0.2, 1.0) * 3870.0                       # Renderer.row_costs: sky rows MISS_COST, ground rows hits
    truth = model * np.where(rows < 700, 3.0, np.where(rows > 1440, 2.2, 1.0)) * (1.0 + 0.1 * np.sin(rows / 97.0))
    measure = lambda bands: [float(truth[a:b].sum()) + 5.0e4 for a, b in bands]     # + a per-band constant (apron rows, launches)
    scale, hist = None, []
    for it in range(5):
        bands = sdist.balanced_row_bands(model * scale if scale is not None else model, world)
        assert bands[0][0] == 0 and bands[-1][1] == H and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
        assert all(a % 4 == 0 for a, _ in bands)
        ms = measure(bands)
        hist.append(max(ms) / (sum(ms) / len(ms)))
        new = sdist.rebalance_scale(model, scale, bands, ms)
        again = sdist.rebalance_scale(model, scale, bands, list(ms))      # same inputs -> same bits (every rank computes this itself)
        assert np.array_equal(new, again) and abs(new.mean() - 1.0) < 1e-12
        scale = new
    assert hist[0] > 1.2 and hist[3] < 1.05 and hist[4] < 1.05, hist
    # garbage measurements leave the scale alone
    s2 = sdist.rebalance_scale(model, scale, bands, [float("nan")] * world)
    assert np.array_equal(s2, scale)
