"""Scene ingestion on the GPU and the compact (uint8 palette) scene volume: SURVEY 8f-4."""
import random

import numpy as np
import pytest
import torch

from conftest import bits

pytestmark = pytest.mark.gpu


def test_ingest_equals_reference_restatement():
    """scene.ingest (csrc/scene.hip kernels) == oracle/pcg_ref.next_world, which tests/test_scene_cpu.py pins on the
    unmodified PCGVoxelGenerator.next_world: block ids (through the palette), camera height map, BEV maps, transform."""
    from oracle import pcg_ref
    from scenedreamer_amd import scene
    for S, seed in ((192, 5), (256, 9)):
        height, sem, tree, models = pcg_ref.synthetic_world(S, seed)
        random.seed(21)
        o = pcg_ref.next_world(height, sem, tree, models)
        random.seed(21)
        sc = scene.ingest(height, sem, tree, models, device="cuda")
        assert sc.voxel_u8.dtype == torch.uint8 and int(sc.palette[0]) == 0
        assert torch.equal(sc.voxel_t.cpu(), o["voxel_t"])
        assert int(((o["voxel_t"] >= 34) & (o["voxel_t"] <= 63)).sum()) > 50
        assert torch.equal(sc.heightmap, o["heightmap"])
        assert torch.equal(sc.current_height_map.cpu(), o["current_height_map"].to(torch.float32))
        assert torch.equal(sc.current_semantic_map.cpu(), o["current_semantic_map"])
        assert torch.equal(sc.trans_mat, o["trans_mat"])


def test_compact_volume_ray_casting_is_bit_identical(oracle):
    """sdn_rvip_u8 on the palette-indexed volume == sdn_rvip on the int32 volume == the oracle, skipping on and off."""
    from scenedreamer_amd import camera, ops, scene, synth
    sc = synth.make_scene(256, 3407)
    vox = sc.voxel_t.cuda()
    u8, pal = scene.compact(vox)
    assert u8.element_size() * 4 == vox.element_size() and torch.equal(pal[u8.long()].to(torch.int32), vox)
    # a strided int32 view compacts to the same thing
    view = vox.permute(2, 0, 1).contiguous().permute(1, 2, 0)
    assert torch.equal(scene.compact(view)[0], u8)
    for (ori, d, up, cf) in camera.eval_camera_poses(sc, maxstep=6):
        f, c, cam_res = camera.frame_intrinsics(cf, (96, 160), 30)
        a = ops.ray_voxel_intersection_perspective(vox, ori, d, up, f, c, cam_res, 6)
        rid, rd2, rrd = oracle.rvip(sc.voxel_t.numpy(), ori.numpy(), d.numpy(), up.numpy(), f, c, cam_res, 6)
        for acc in (True, False):
            b = ops.ray_voxel_intersection_perspective(u8, ori, d, up, f, c, cam_res, 6, accelerate=acc, palette=pal)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))
            assert torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
            np.testing.assert_array_equal(b[0].cpu().numpy(), rid)
            np.testing.assert_array_equal(bits(b[1].cpu().numpy()), bits(rd2))


def test_renderer_on_compact_scene_is_bit_identical(weights_full):
    """Frames rendered from an ingested CompactScene == frames rendered from its expanded int32 volume."""
    from oracle import pcg_ref
    from scenedreamer_amd import camera, scene, synth
    from scenedreamer_amd.renderer import Renderer
    height, sem, tree, models = pcg_ref.synthetic_world(256, 2)
    random.seed(3)
    sc = scene.ingest(height, sem, tree, models, device="cuda")
    plain = synth.Scene()
    plain.voxel_t = sc.voxel_t.clone()
    plain.heightmap, plain.current_height_map, plain.current_semantic_map = sc.heightmap, sc.current_height_map, sc.current_semantic_map
    plain.trans_mat, plain.sample_size = sc.trans_mat, sc.sample_size
    imgs = []
    for s in (sc, plain):
        R = Renderer(weights_full, s, "cuda")
        R.set_style(synth.make_style(8888))
        pose = camera.eval_camera_poses(s, maxstep=8)[2]
        imgs.append(R.render_frame(pose, (72, 96), 12, mode="fused").clone())
    assert torch.equal(imgs[0], imgs[1]) and float(imgs[0].std()) > 0.01
