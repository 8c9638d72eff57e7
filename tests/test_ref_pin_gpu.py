"""HIP ops against the REFERENCE'S OWN native sources compiled for the host (oracle/_ref, prebuilt in the build
container by oracle/build_ref.py and shipped with the snapshot): no restatement in between."""
import numpy as np
import pytest
import torch

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    from oracle import build_ref as BR
    if not BR.built("nofma"):
        pytest.skip("oracle/_ref not present in this snapshot")
    from oracle import ref_native as R
    return R


@pytest.fixture(scope="module")
def ops():
    from scenedreamer_amd import capi, ops
    capi.lib()
    return ops


@pytest.mark.parametrize("S,seed", [(256, 3407), (512, 11)])
def test_rvip_hip_equals_reference_source(ops, ref, S, seed):
    from scenedreamer_amd import camera, synth
    sc = synth.make_scene(S, seed)
    vox_np, vox_dev = sc.voxel_t.numpy(), sc.voxel_t.cuda()
    for (ori, d, up, cf) in camera.eval_camera_poses(sc, maxstep=8):
        for hw in ((96, 160), (33, 61)):
            f, c, cam_res = camera.frame_intrinsics(cf, hw, 30)
            rid, rd2, rrd = ref.rvip(vox_np, ori.numpy(), d.numpy(), up.numpy(), f, c, cam_res, 6)
            vid, d2, rd = ops.ray_voxel_intersection_perspective(vox_dev, ori, d, up, f, c, cam_res, 6)
            np.testing.assert_array_equal(vid.cpu().numpy(), rid)
            np.testing.assert_array_equal(bits(d2.cpu().numpy()), bits(rd2))
            np.testing.assert_array_equal(bits(rd.cpu().numpy()), bits(rrd))


@pytest.mark.parametrize("case", [(5, 8, 16, 16, 2 ** (7 / 15), 19, 0, False), (3, 2, 8, 4, 1.7, 12, 0, False),
                                  (2, 4, 6, 4, 1.5, 14, 0, True), (4, 1, 5, 8, 2.0, 10, 1, False)])
def test_grid_encode_hip_equals_reference_source(ops, ref, case):
    from scenedreamer_amd.gridencoder import level_offsets
    D, C, L, H, pls, T, gridtype, ac = case
    rng = np.random.default_rng(D * 100 + C)
    offs = level_offsets(D, L, pls, H, T, ac)
    emb = rng.random((int(offs[-1]), C), dtype=np.float32) - 0.5
    x = rng.random((5000, D), dtype=np.float32)
    x[:8] = rng.random((8, D), dtype=np.float32) * 1.2 - 0.1
    S = np.float32(np.log2(pls))
    want, want_dd = ref.grid_encode_fwd(x, emb, offs, S, H, True, gridtype, ac)
    B = x.shape[0]
    out = torch.empty(L, B, C, device="cuda")
    dd = torch.empty(B, L * D * C, device="cuda")
    ops.grid_encode_forward(torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(),
                            torch.from_numpy(np.asarray(offs, np.int32)).cuda(), out, B, D, C, L, float(S), H, True, dd,
                            gridtype, ac)
    # same products, different summation order over the 2^D corners: 1e-5
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=1e-5)
    np.testing.assert_allclose(dd.cpu().numpy(), want_dd, rtol=1e-4, atol=1e-4 * float(np.abs(want_dd).max()))


def test_posenc_hip_equals_reference_source(ops, ref):
    x = (np.random.default_rng(0).random((1, 37, 53, 1, 3), dtype=np.float32) * 2 - 1)
    y = ops.positional_encoding(torch.from_numpy(x).cuda(), 5, -1, True)
    np.testing.assert_allclose(y.cpu().numpy(), ref.posenc_fwd(x, 5, -1, True), rtol=1e-5, atol=1e-5)
