"""Parity of the three drop-in HIP ops against the CPU oracle (runs on the MI355X box).

Every call goes through the C ABI (scenedreamer_amd.ops -> ctypes -> libsdnative.so).
Integer / index outputs and the ray marcher's float outputs must be BIT-EXACT;
the trigonometric and gather ops are compared with the tolerance written in each test.
"""
import numpy as np
import pytest
import torch

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from scenedreamer_amd import capi, ops
    capi.lib()  # fail loudly if the HIP library is missing
    assert torch.cuda.is_available()
    return ops


def _rvip_both(ops, O, vox_np, pose, f, c, dims, M, vox_dev=None):
    ori, d, up = [np.asarray(x, np.float32) for x in pose]
    if vox_dev is None:
        vox_dev = torch.from_numpy(vox_np).cuda()
    rid, rd2, rrd = O.rvip(vox_np, ori, d, up, f, c, dims, M)
    for accelerate in (True, False):   # exact empty-space skipping on / off: the same bits
        vid, d2, rd = ops.ray_voxel_intersection_perspective(vox_dev, torch.from_numpy(ori), torch.from_numpy(d),
                                                             torch.from_numpy(up), f, c, dims, M, accelerate=accelerate)
        torch.cuda.synchronize()
        assert vid.shape == (dims[0], dims[1], M, 1) and d2.shape == (2, dims[0], dims[1], M, 1)
        assert rd.shape == (dims[0], dims[1], 1, 3)
        np.testing.assert_array_equal(vid.cpu().numpy(), rid)
        np.testing.assert_array_equal(bits(d2.cpu().numpy()), bits(rd2))
        np.testing.assert_array_equal(bits(rd.cpu().numpy()), bits(rrd))
    return rid


@pytest.mark.parametrize("seed", [3407, 11, 12])
def test_rvip_bit_exact_orbit(ops, oracle, seed):
    from scenedreamer_amd import camera, synth
    sc = synth.make_scene(256, seed)
    vox_np = sc.voxel_t.numpy()
    vox_dev = sc.voxel_t.cuda()
    poses = camera.eval_camera_poses(sc, maxstep=10)
    nhit = 0
    for i, (ori, d, up, cf) in enumerate(poses):
        for hw in ((128, 128), (33, 61)):
            f, c, cam_res = camera.frame_intrinsics(cf, hw, 30)
            rid = _rvip_both(ops, oracle, vox_np, (ori.numpy(), d.numpy(), up.numpy()), f, c, cam_res, 6, vox_dev)
            nhit += int((rid != 0).sum())
    assert nhit > 0


def test_rvip_edge_cases(ops, oracle):
    vox = np.zeros((8, 16, 16), np.int32)
    vox[2:5, 4:12, 4:12] = 9
    vox[6, 8, 8] = 680 - 1
    up = [1.0, 0.0, 0.0]
    cases = [
        ([4.5, 8.5, -5.5], [0, 0, 1], 1),          # axis-aligned ray from outside the volume (dir has exact zeros)
        ([7.5, 8.5, 8.5], [-1, 0.0, 0.0], 3),      # straight down through the stack, up parallel to dir -> NaN frame
        ([20.0, 30.0, 40.0], [1, 1, 1], 6),         # pointing away: every ray misses
        ([3.0, 8.0, 8.0], [0.2, 1, 0.3], 6),        # origin on integer coordinates, inside a solid voxel
        ([7.999, 0.001, 15.999], [-0.3, 0.7, -0.6], 2),
    ]
    for ori, d, M in cases:
        for up_ in (up, [0.0, 1.0, 0.0]):
            _rvip_both(ops, oracle, vox, (ori, d, up_), 20.0, [7.5, 9.5], [16, 20], M)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_rvip_sparse_volume_block_skipping(ops, oracle, seed):
    """Mostly empty volumes (floating cells, thin sheets, extents that are not multiples of the 8x16x16 occupancy
    block) from random poses inside, on the boundary of and outside the volume: long empty-block jumps must land
    in exactly the state of the cell-by-cell walk."""
    rng = np.random.default_rng(seed)
    dims = [(37, 150, 131), (64, 96, 200), (9, 257, 63)][seed - 1]
    vox = np.zeros(dims, np.int32)
    n = 60
    vox[rng.integers(0, dims[0], n), rng.integers(0, dims[1], n), rng.integers(0, dims[2], n)] = rng.integers(1, 50, n)
    vox[dims[0] // 2, :, dims[2] // 3] = 7                       # a line
    vox[0, 10:dims[1] - 10, 5:dims[2] - 5] = 3                   # a floor sheet
    vox_dev = torch.from_numpy(vox).cuda()
    nhit = 0
    for k in range(12):
        if k % 3 == 0:      # outside, looking at the centre
            ori = np.array([dims[0] * 1.5, -20.0 - k, dims[2] * 0.5 + k], np.float32)
        elif k % 3 == 1:    # inside
            ori = (rng.random(3) * np.array(dims)).astype(np.float32)
        else:               # exactly on integer planes / the boundary
            ori = np.array([float(dims[0]), float(rng.integers(0, dims[1])), float(rng.integers(0, dims[2]))], np.float32)
        d = (np.array(dims, np.float32) * rng.random(3).astype(np.float32) - ori)
        if k == 7:
            d = np.array([0.0, 1.0, 0.0], np.float32)             # exact zeros in the direction
        rid = _rvip_both(ops, oracle, vox, (ori, d, [1.0, 0.0, 0.0]), 35.0, [23.5, 31.5], [48, 64], 5, vox_dev)
        nhit += int((rid != 0).sum())
    assert nhit > 0


def test_rvip_occupancy_follows_volume_edits(ops, oracle):
    vox = np.zeros((16, 64, 64), np.int32)
    vox[3, 20:40, 20:40] = 5
    dev = torch.from_numpy(vox).cuda()
    pose = ([14.5, 5.5, 5.5], [-0.4, 0.7, 0.6], [1.0, 0.0, 0.0])
    _rvip_both(ops, oracle, vox, pose, 30.0, [15.5, 15.5], [32, 32], 3, dev)
    vox[9, 8:12, 8:12] = 11                                       # in-place edit: the cached grid must be rebuilt
    dev[9, 8:12, 8:12] = 11
    _rvip_both(ops, oracle, vox, pose, 30.0, [15.5, 15.5], [32, 32], 3, dev)


def test_rvip_strided_volume(ops, oracle):
    from scenedreamer_amd import synth
    sc = synth.make_scene(64, 5)
    base = sc.voxel_t.permute(2, 0, 1).contiguous()           # memory order z, x, y
    view = base.permute(1, 2, 0)                               # logical x, y, z with odd strides
    assert not view.is_contiguous()
    vox_np = np.ascontiguousarray(view.numpy())
    pose = ([30.0, 10.2, 60.7], [-0.5, 0.6, -0.8], [1.0, 0, 0])
    ori, d, up = [np.asarray(x, np.float32) for x in pose]
    vid, d2, rd = ops.ray_voxel_intersection_perspective(view.cuda().permute(2, 0, 1).contiguous().permute(1, 2, 0),
                                                         torch.from_numpy(ori), torch.from_numpy(d),
                                                         torch.from_numpy(up), 40.0, [23.5, 31.5], [48, 64], 4)
    rid, rd2, _ = oracle.rvip(vox_np, ori, d, up, 40.0, [23.5, 31.5], [48, 64], 4)
    np.testing.assert_array_equal(vid.cpu().numpy(), rid)
    np.testing.assert_array_equal(bits(d2.cpu().numpy()), bits(rd2))


def test_rvip_rejects_bad_input(ops):
    with pytest.raises(RuntimeError):
        ops.ray_voxel_intersection_perspective(torch.zeros(4, 4, 4, dtype=torch.int32), torch.zeros(3), torch.ones(3),
                                               torch.ones(3), 1.0, [0, 0], [4, 4], 2)
    with pytest.raises(RuntimeError):
        ops.ray_voxel_intersection_perspective(torch.zeros(4, 4, 4, dtype=torch.float32).cuda(), torch.zeros(3),
                                               torch.ones(3), torch.ones(3), 1.0, [0, 0], [4, 4], 2)


@pytest.mark.parametrize("shape,dim,ndeg,incl", [((1, 37, 53, 1, 3), -1, 5, True), ((64, 7), -1, 4, False),
                                                 ((5, 6, 7), 1, 3, True), ((3, 1), 0, 1, True)])
def test_posenc_forward_backward(ops, oracle, shape, dim, ndeg, incl):
    rng = np.random.default_rng(0)
    x = (rng.random(shape, dtype=np.float32) * 2 - 1)
    y = ops.positional_encoding(torch.from_numpy(x).cuda(), ndeg, dim, incl)
    ref = oracle.posenc_fwd(x, ndeg, dim, incl)
    assert tuple(y.shape) == ref.shape
    # same tolerance as the reference's own self check (positional_encoding.py:57-63)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    # the reference's pure-torch twin, positional_encoding_pt (positional_encoding.py:45-54)
    xt = torch.from_numpy(x)
    twin = torch.cat([fn(xt * np.pi * 2 ** i) for i in range(ndeg) for fn in (torch.sin, torch.cos)] +
                     ([xt] if incl else []), dim=dim)
    np.testing.assert_allclose(y.cpu().numpy(), twin.numpy(), rtol=1e-5, atol=2e-5)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    gi = ops.positional_encoding_backward(torch.from_numpy(g).cuda(), y, ndeg, dim, incl)
    np.testing.assert_allclose(gi.cpu().numpy(), oracle.posenc_bwd(g, ref, ndeg, dim, incl), rtol=1e-4, atol=1e-4)


def test_posenc_large_arguments(ops, oracle):
    # the reference's self check uses inputs up to 1024 (positional_encoding.py:58)
    x = (np.random.default_rng(1).random((64, 48), dtype=np.float32) * 1024)
    y = ops.positional_encoding(torch.from_numpy(x).cuda(), 4, -1, True)
    np.testing.assert_allclose(y.cpu().numpy(), oracle.posenc_fwd(x, 4, -1, True), rtol=1e-5, atol=1e-5)


def _grid_case(D, C, L, log2_T, base, pls, gridtype, align, B, seed):
    from scenedreamer_amd.gridencoder import level_offsets
    rng = np.random.default_rng(seed)
    offs = level_offsets(D, L, pls, base, log2_T, align)
    emb = (rng.random((int(offs[-1]), C), dtype=np.float32) - 0.5)
    x = rng.random((B, D), dtype=np.float32)
    x[::17] = 0.0
    x[5::23] = 1.0
    x[3::31, 0] = -0.01      # out of range -> zeros
    x[7::29, D - 1] = 1.001
    return offs, emb, x, np.float32(np.log2(pls))


@pytest.mark.parametrize("D", [2, 3, 4, 5])
@pytest.mark.parametrize("C", [1, 2, 4, 8])
def test_grid_forward_all_instantiations(ops, oracle, D, C):
    for gridtype, align in ((0, False), (1, False), (0, True)):
        offs, emb, x, S = _grid_case(D, C, 6, 11, 4, 1.7, gridtype, align, 3001, D * 10 + C)
        L = offs.size - 1
        out = torch.empty(L, x.shape[0], C, device="cuda")
        dy = torch.empty(x.shape[0], L * D * C, device="cuda")
        ops.grid_encode_forward(torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(),
                                torch.from_numpy(offs).cuda(), out, x.shape[0], D, C, L, S, 4, True, dy, gridtype, align)
        ref, ref_dy = oracle.grid_encode_fwd(x, emb, offs, S, 4, True, gridtype, align)
        # tolerance: fp32 reassociation only (hipcc may contract w*v+acc into FMA), |emb| <= 0.5
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=1e-5)
        np.testing.assert_allclose(dy.cpu().numpy(), ref_dy, rtol=1e-4, atol=2e-3)   # dy_dx carries a factor `scale` <= ~60
        oob = (x < 0).any(1) | (x > 1).any(1)
        assert oob.any() and (out.cpu().numpy()[:, oob] == 0).all()


def test_grid_forward_scenedreamer_config(ops, oracle):
    """D=5, C=8, L=16, T=2^19: the instance SceneDreamer builds (scenedreamer.py:51)."""
    from scenedreamer_amd import synth
    w = synth.make_weights(0)
    offs, emb = w["hash_encoder.offsets"], w["hash_encoder.embeddings"]
    rng = np.random.default_rng(3)
    B = 20000
    x = rng.random((B, 5), dtype=np.float32)
    x[:, 3] = 0.36
    x[:, 4] = 0.63
    x[::101, 1] = 1.5
    S = np.float32(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    out = torch.empty(16, B, 8, device="cuda")
    dy = torch.empty(1, device="cuda")
    ops.grid_encode_forward(torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(), torch.from_numpy(offs).cuda(),
                            out, B, 5, 8, 16, S, 16, False, dy, 0, False)
    np.testing.assert_allclose(out.cpu().numpy(), oracle.grid_encode_fwd(x, emb, offs, S, 16), rtol=0, atol=1e-5)


def test_grid_forward_half_table(ops, oracle):
    """scalar_t = at::Half (gridencoder.cu:140-176, :181-223): `results` / `results_grad` are halves, every product and sum
    rounds to half.  Features and dy_dx bit for bit against the oracle's emulation, which is pinned on the reference's own
    kernel (tests/test_ref_pin_cpu.py)."""
    for (D, C, L, T, H, pls, gt, ac, B, seed) in ((3, 4, 5, 10, 4, 1.6, 0, False, 1500, 77), (5, 8, 4, 14, 16, 1.4, 0, False, 900, 5),
                                                  (2, 2, 3, 10, 4, 1.5, 1, True, 700, 6)):
        offs, emb, x, S = _grid_case(D, C, L, T, H, pls, gt, ac, B, seed)
        x[::40] = 1.2                                       # out-of-range rows
        emb16 = torch.from_numpy(emb).half()
        out = torch.empty(L, B, C, device="cuda", dtype=torch.half)
        dy = torch.empty(B, L * D * C, device="cuda", dtype=torch.half)
        ops.grid_encode_forward(torch.from_numpy(x).cuda(), emb16.cuda(), torch.from_numpy(offs).cuda(), out, B,
                                D, C, L, S, H, True, dy, gt, ac)
        ref, ref_dy = oracle.grid_encode_fwd_f16(x, emb16.numpy(), offs, S, H, True, gt, ac)
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint16), ref.view(np.uint16))
        np.testing.assert_array_equal(dy.cpu().numpy().view(np.uint16), ref_dy.view(np.uint16))


def test_grid_backward(ops, oracle):
    for D, C in ((3, 2), (5, 8), (2, 1), (4, 4)):
        offs, emb, x, S = _grid_case(D, C, 4, 9, 4, 1.5, 0, False, 700, D + C)
        L = offs.size - 1
        B = x.shape[0]
        rng = np.random.default_rng(5)
        grad = rng.standard_normal((L, B, C)).astype(np.float32)
        xd, ed, od = torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(), torch.from_numpy(offs).cuda()
        out = torch.empty(L, B, C, device="cuda")
        dy = torch.empty(B, L * D * C, device="cuda")
        ops.grid_encode_forward(xd, ed, od, out, B, D, C, L, S, 4, True, dy, 0, False)
        gg = torch.zeros_like(ed)
        gi = torch.zeros(B, D, device="cuda")
        ops.grid_encode_backward(torch.from_numpy(grad).cuda(), xd, ed, od, gg, B, D, C, L, S, 4, True, dy, gi, 0, False)
        _, ref_dy = oracle.grid_encode_fwd(x, emb, offs, S, 4, True)
        rgg, rgi = oracle.grid_encode_bwd(grad, x, emb.shape, offs, S, 4, ref_dy)
        # atomics accumulate in arbitrary order: tolerance scales with the number of colliding samples
        np.testing.assert_allclose(gg.cpu().numpy(), rgg, rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(gi.cpu().numpy(), rgi, rtol=1e-4, atol=5e-3)


def test_grid_backward_half(ops, oracle):
    """f16 tables / gradients (gridencoder.cu:296-304: contributions rounded to half, packed half2 atomics; :317-343:
    the input gradient accumulated sequentially in half).  grad_inputs must equal the oracle's sequential half
    evaluation BIT FOR BIT; the table gradient is a half-precision sum in arbitrary order: tolerance ~ a few half
    ulps of the largest entry.  C = 1 (odd channel count: the reference's at::Half atomicAdd is an empty stub) goes
    through the CAS path and is checked the same way."""
    for D, C in ((3, 2), (5, 8), (2, 1), (4, 4)):
        offs, emb, x, S = _grid_case(D, C, 4, 9, 4, 1.5, 0, False, 500, 3 * D + C)
        L = offs.size - 1
        B = x.shape[0]
        rng = np.random.default_rng(9)
        grad = rng.standard_normal((L, B, C)).astype(np.float16)
        xd, od = torch.from_numpy(x).cuda(), torch.from_numpy(offs).cuda()
        ed = torch.from_numpy(emb).half().cuda()
        out = torch.empty(L, B, C, device="cuda", dtype=torch.half)
        dy = torch.empty(B, L * D * C, device="cuda", dtype=torch.half)
        ops.grid_encode_forward(xd, ed, od, out, B, D, C, L, S, 4, True, dy, 0, False)
        gg = torch.zeros_like(ed)
        gi = torch.zeros(B, D, device="cuda", dtype=torch.half)
        ops.grid_encode_backward(torch.from_numpy(grad).cuda(), xd, ed, od, gg, B, D, C, L, S, 4, True, dy, gi, 0, False)
        rgg, rgi = oracle.grid_encode_bwd_f16(grad, x, emb.shape, offs, S, 4, dy.cpu().numpy())
        np.testing.assert_array_equal(gi.cpu().numpy().view(np.uint16), rgi.view(np.uint16))
        scale = float(np.abs(rgg).max())
        np.testing.assert_allclose(gg.float().cpu().numpy(), rgg, rtol=0, atol=1e-2 * scale)
        assert float(np.abs(rgg).max()) > 0.5


def test_grid_rejects_unsupported(ops):
    x = torch.rand(8, 3, device="cuda")
    offs = torch.tensor([0, 64], dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="C must be"):
        ops.grid_encode_forward(x, torch.rand(64, 3, device="cuda"), offs, torch.empty(1, 8, 3, device="cuda"), 8, 3, 3,
                                1, 0.0, 4, False, torch.empty(1, device="cuda"), 0, False)
    with pytest.raises(RuntimeError, match="D must be"):
        ops.grid_encode_forward(torch.rand(8, 6, device="cuda"), torch.rand(64, 2, device="cuda"), offs,
                                torch.empty(1, 8, 2, device="cuda"), 8, 6, 2, 1, 0.0, 4, False,
                                torch.empty(1, device="cuda"), 0, False)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.grid_encode_forward(x.cpu(), torch.rand(64, 2), offs.cpu(), torch.empty(1, 8, 2), 8, 3, 2, 1, 0.0, 4, False,
                                torch.empty(1), 0, False)


def test_gridencoder_module_and_autograd(ops, oracle):
    from scenedreamer_amd.gridencoder import GridEncoder
    torch.manual_seed(0)
    enc = GridEncoder(input_dim=3, num_levels=4, level_dim=2, base_resolution=4, log2_hashmap_size=9,
                      desired_resolution=32).cuda()
    enc.embeddings.data.uniform_(-0.5, 0.5)
    x = (torch.rand(2, 50, 3, device="cuda") * 2 - 1).requires_grad_(True)
    y = enc(x)
    assert y.shape == (2, 50, 8)
    S = np.float32(np.log2(enc.per_level_scale))
    ref = oracle.grid_encode_fwd(((x.detach().cpu().numpy().reshape(-1, 3) + 1) / 2).astype(np.float32),
                                 enc.embeddings.detach().cpu().numpy(), enc.offsets.cpu().numpy(), S, 4)
    np.testing.assert_allclose(y.detach().cpu().numpy().reshape(-1, 4, 2).transpose(1, 0, 2), ref, atol=1e-5)
    y.square().sum().backward()
    assert enc.embeddings.grad is not None and torch.isfinite(enc.embeddings.grad).all()
    assert x.grad is not None and x.grad.abs().sum() > 0
