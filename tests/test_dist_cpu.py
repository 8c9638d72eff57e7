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
