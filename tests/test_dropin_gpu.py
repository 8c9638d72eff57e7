"""The fast path behind the reference's own surface, on the MI355X (scenedreamer_amd/modules.py, dropin.py).

Module level: LightningMLP / SKYMLP / RenderCNN with the reference's constructor, parameters and forward signature, forward
on the MFMA kernels (sdn_render_mlp, sdn_sky_mlp, MfmaCNN) -- against their own composite (plain PyTorch fp32) forward on
the same weights, which tests/test_dropin_cpu.py pins on the reference's classes.
Generator level: the UNMODIFIED imaginaire generator with `install_shims(fast=True)` -- _forward_perpix / _forward_global on
the goldens, and its own inference_givenstyle frame loop against this package's renderer."""
import os

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _load(module, weights, prefix):
    sd = {k[len(prefix):]: torch.as_tensor(np.asarray(v)) for k, v in weights.items() if k.startswith(prefix)}
    module.load_state_dict(sd)
    return module.cuda().eval()


@pytest.fixture(scope="module")
def nets(weights_full):
    from scenedreamer_amd import modules
    mlp = _load(modules.LightningMLP(128, 256, 0, mask_dim=12, out_channels_s=1, out_channels_c=64), weights_full, "render_net.")
    sky = _load(modules.SKYMLP(33, 256, 64), weights_full, "sky_net.")
    cnn = _load(modules.RenderCNN(64, 256), weights_full, "denoiser.")
    for m in (mlp, sky, cnn):
        for p in m.parameters():
            p.requires_grad_(False)          # as inference.py does (:63-64): nothing asks for gradients
    return mlp, sky, cnn


def _style_code(seed=0):
    g = golden("style_globalenc.npz")
    z = torch.from_numpy(g["z"]).cuda().reshape(1, -1)
    if seed:
        z = z + 0.25 * torch.randn(z.shape, generator=torch.Generator().manual_seed(seed)).cuda()
    return z


@pytest.mark.parametrize("shape", [(1, 7, 9, 5), (1, 40, 56, 12), (2, 3, 5, 4), (1, 1, 1, 1)])
def test_lightning_mlp_module_on_the_mfma_kernel(nets, shape):
    """LightningMLP.forward (layers.py:92-126) -> sdn_render_mlp, ragged row counts (not multiples of 32 / 256), batch 2."""
    mlp = nets[0]
    n, h, w, ms = shape
    gen = torch.Generator().manual_seed(h * w)
    x = (torch.rand((n, h, w, ms, 128), generator=gen) - 0.5).cuda() * 0.6          # hash-grid features: |x| <= ~0.5 x 8 corners' blend
    lab = torch.randint(0, 12, (n, h, w, ms, 1), generator=gen).cuda()
    m = torch.zeros((n, h, w, ms, 12), device="cuda").scatter_(-1, lab, 1.0)
    z = torch.cat([_style_code(i) for i in range(n)], dim=0)
    with torch.no_grad():
        sigma, c = mlp(x, None, z, m)
        assert mlp.native_reason(x, None, z, m) is None
        ref_s, ref_c = mlp._forward_composite(x, None, z, m)
    assert sigma.shape == ref_s.shape == (n, h, w, ms, 1) and c.shape == ref_c.shape == (n, h, w, ms, 64)
    es = float((sigma - ref_s).abs().max() / max(1.0, float(ref_s.abs().max())))
    ec = float((c - ref_c).abs().max())
    print(f"LightningMLP {shape}: sigma rel err {es:.2e} (|sigma| <= {float(ref_s.abs().max()):.2f}), colour abs err {ec:.2e}")
    assert es < 2e-4 and ec < 2e-4


def test_lightning_mlp_notices_new_weights_and_styles(nets):
    """The packed weights follow in-place parameter updates and a changed style code (Backend.bind / style)."""
    from scenedreamer_amd import modules
    mlp = nets[0]
    mine = modules.LightningMLP(128, 256, 0, mask_dim=12, out_channels_s=1, out_channels_c=64)
    mine.load_state_dict(mlp.state_dict())
    mine = mine.cuda().eval()
    for p in mine.parameters():
        p.requires_grad_(False)
    x = (torch.rand((1, 5, 6, 4, 128), generator=torch.Generator().manual_seed(3)) - 0.5).cuda()
    m = torch.zeros((1, 5, 6, 4, 12), device="cuda")
    m[..., 4] = 1
    z = _style_code()
    with torch.no_grad():
        a = mine(x, None, z, m)
        mine.fc_3.weight.mul_(1.5)
        mine.fc_1.bias.add_(0.05)
        b = mine(x, None, z, m)
        rb = mine._forward_composite(x, None, z, m)
        z.mul_(0.5)
        c = mine(x, None, z, m)
        rc = mine._forward_composite(x, None, z, m)
    assert float((a[1] - b[1]).abs().max()) > 1e-3
    assert float((b[1] - rb[1]).abs().max()) < 2e-4 and float((c[1] - rc[1]).abs().max()) < 2e-4


def test_a_new_style_tensor_on_a_recycled_address_is_folded_again(nets):
    """the reference builds z = style_net(style) afresh per call and frees the old one; the caching allocator
    hands the next z the same address with version counter 0.  The backends hold the tensor their fold was keyed on, so the new
    style cannot be mistaken for it: all three networks render the NEW style, and one-hot-ness of `m` decides native vs
    composite."""
    mlp, sky, cnn = nets
    x = (torch.rand((1, 5, 6, 4, 128), generator=torch.Generator().manual_seed(3)) - 0.5).cuda()
    m = torch.zeros((1, 5, 6, 4, 12), device="cuda")
    m[..., 7] = 1
    rows = (torch.rand((1, 50, 33), generator=torch.Generator().manual_seed(4)) * 2 - 1).cuda()
    img_in = (torch.rand((1, 64, 24, 40), generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()
    outs = []
    with torch.no_grad():
        for seed in (0, 1):
            z = _style_code(seed)                   # a fresh tensor per "call of the generator"
            o = (mlp(x, None, z, m)[1], sky(rows, z), cnn(img_in, z))
            r = (mlp._forward_composite(x, None, z, m)[1], sky._forward_composite(rows, z), cnn._forward_composite(img_in, z))
            for a, b, tol in zip(o, r, (2e-4, 3e-4, 5e-3)):
                assert float((a - b).abs().max()) < tol * max(1.0, float(b.abs().max())), seed
            outs.append(o)
            del z
            torch.cuda.synchronize()
        assert all(float((a - b).abs().max()) > 1e-3 for a, b in zip(*outs))     # the second style was rendered, not the first again
        # soft labels are not what the label-bias shortcut computes: composite path, same answer as the reference arithmetic
        z = _style_code(1)
        soft = m * 0.9 + 0.1 / 12
        assert mlp.native_reason(x, None, z, m) is None and mlp.native_reason(x, None, z, soft) == "m is not one-hot"
        a, b = mlp(x, None, z, soft), mlp._forward_composite(x, None, z, soft)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_trunk_weights_beyond_f16_fall_back_instead_of_raising(nets):
    """a style whose scaled trunk weights leave f16's range is served by the composite forward (the explicit
    Renderer API still raises fused.TrunkRangeError)."""
    from scenedreamer_amd import modules
    mlp = nets[0]
    mine = modules.LightningMLP(128, 256, 0, mask_dim=12, out_channels_s=1, out_channels_c=64)
    mine.load_state_dict(mlp.state_dict())
    mine = mine.cuda().eval()
    for p in mine.parameters():
        p.requires_grad_(False)
    x = (torch.rand((1, 3, 4, 2, 128), generator=torch.Generator().manual_seed(3)) - 0.5).cuda()
    m = torch.zeros((1, 3, 4, 2, 12), device="cuda")
    m[..., 2] = 1
    z = _style_code()
    with torch.no_grad():
        mine.fc_1.weight.mul_(2000.0)               # |w| * 2^8 beyond 32768
        a = mine(x, None, z, m)
        b = mine._forward_composite(x, None, z, m)
    assert "f16" in mine.__dict__.get("_sdn_composite_reason", "") and torch.equal(a[1], b[1])


def test_sky_mlp_module_encoded_rows_and_tagged_directions(nets):
    """SKYMLP.forward (gancraft_base.py:150-169): (a) any [.., 33] rows -> sky_kernel<PRE>; (b) the output of this package's
    voxlib.positional_encoding -> the kernel encodes the ray directions itself; (c) an in-place edit of that tensor voids the tag."""
    from scenedreamer_amd import ops
    sky = nets[1]
    z = _style_code()
    gen = torch.Generator().manual_seed(5)
    rd = torch.nn.functional.normalize(torch.randn((1, 37, 53, 1, 3), generator=gen), dim=-1).cuda()
    with torch.no_grad():
        pe = ops.positional_encoding(rd.expand(-1, -1, -1, 1, -1).contiguous(), 5, -1, True)
        ref = sky._forward_composite(pe, z)
        tagged = sky(pe, z)                          # (b)
        assert "_sdn_last_frame" in sky.__dict__ and sky.__dict__["_sdn_last_frame"]["n_rays"] == 37 * 53
        plain = sky(pe.clone(), z)                   # (a): a copy carries no tag
        rows = (torch.rand((1, 11, 33), generator=gen) * 2 - 1).cuda()
        r2, ref2 = sky(rows, z), sky._forward_composite(rows, z)
        pe2 = ops.positional_encoding(rd.contiguous(), 5, -1, True)
        pe2[..., 3:6] += 0.25                        # (c) no longer the encoding of rd
        edited, ref3 = sky(pe2, z), sky._forward_composite(pe2, z)
    assert tagged.shape == ref.shape == (1, 37, 53, 1, 64)
    for name, a, b in (("tagged", tagged, ref), ("encoded rows", plain, ref), ("arbitrary rows", r2, ref2), ("edited", edited, ref3)):
        e = float((a - b).abs().max())
        print(f"SKYMLP {name}: max abs err {e:.2e}")
        assert e < 3e-4, name
    assert float((edited - tagged).abs().max()) > 1e-3


@pytest.mark.parametrize("hw", [(40, 56), (33, 71)])
def test_render_cnn_module_returns_the_pre_tanh_image(nets, hw):
    """RenderCNN.forward (gancraft_base.py:202-225) returns conv4's output; _forward_global applies tanh (:598-603)."""
    cnn = nets[2]
    z = _style_code()
    x = (torch.rand((1, 64, hw[0], hw[1]), generator=torch.Generator().manual_seed(9)) * 2 - 1).cuda()
    with torch.no_grad():
        raw = cnn(x, z)
        ref = cnn._forward_composite(x, z)
    assert cnn.native_reason(x, z) is None and raw.shape == ref.shape == (1, 3, hw[0], hw[1])
    e = float((raw - ref).abs().max())
    et = float((torch.tanh(raw) - torch.tanh(ref)).abs().max())
    print(f"RenderCNN {hw}: pre-tanh max abs err {e:.2e} (|y| <= {float(ref.abs().max()):.2f}), after tanh {et:.2e}")
    assert et < TOL and e < 5e-3 * max(1.0, float(ref.abs().max()))


def test_field_render_aux_outputs_and_device_camera(weights_full, scene256):
    """sdn_field_render's additions: cam_ori read from device memory == host values (bitwise); depth_out == the sample
    depths of the stand-alone sampling op (the same device function); weights_out: zero for rays without a hit, non-negative,
    summing to at most 1 per ray (the generator-level test compares them with the reference's own method)."""
    from scenedreamer_amd import camera, fused, ops, synth
    from scenedreamer_amd.renderer import Renderer
    R = Renderer(weights_full, scene256, "cuda")
    R.set_style(synth.make_style(8888))
    pose = camera.eval_camera_poses(scene256, maxstep=8)[2]
    ns = 12
    with torch.no_grad():
        vid, d2, rd, (H0, W0) = R.cast_rays(pose, (48, 64))
        n = H0 * W0
        vid, d2, rd = vid.view(n, R.M), d2.view(2, n, R.M), rd.view(n, 3)
        sky_c, sky_avg = fused.sky_fused(R, rd)
        host = fused.field_render(R, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, ns)
        dev = fused.field_render(R, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32).cuda(), sky_c, sky_avg, ns)
        aux = dict.fromkeys(fused.AUX_OUTPUTS)
        with_aux = fused.field_render(R, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, ns, aux=aux)
        two = {}                 # (an empty dict asks for weights + depth only)
        fused.field_render(R, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, ns, aux=two)
        depth, _, _ = ops.sample_depth_batched(d2.view(1, 2, H0, W0, R.M, 1), ns + 1, deterministic=True, use_box_boundaries=False,
                                               sample_depth=R.sample_depth)
        depth = depth.view(n, ns)
        depth = torch.where(torch.isnan(depth) | torch.isinf(depth), torch.zeros_like(depth), depth)
    assert torch.equal(host, dev)
    # (the instantiation that also stores the weights is another compilation of the same arithmetic: hipcc contracts
    # w * rgb + acc into an fma or not depending on the other uses of w -- not the same bits, the same value to an ulp)
    assert float((host - with_aux).abs().max()) < 2e-6
    w, dp = aux["weights"], aux["depth"]
    assert tuple(w.shape) == tuple(dp.shape) == (n, ns)
    hit = vid[:, 0] != 0
    assert float(w[~hit].abs().max()) == 0.0 and bool((w >= 0).all()) and float(w.sum(dim=1).max()) <= 1.0 + 1e-5
    assert int(hit.sum()) > 100 and float(w[hit].sum(dim=1).max()) > 0.05
    assert torch.equal(dp, depth) and sorted(two) == ["depth", "weights"] and torch.equal(two["weights"], w) and torch.equal(two["depth"], dp)
    # the volume term of net_out recomposed from the per-sample outputs: sum_k w_k (clamp(c_k) + 1) + (1 - T)(clamp(sky) + 1) - 1
    sig, col, skyb, nosky = aux["sigma"], aux["colour"], aux["sky_blended"], aux["nosky"]
    assert tuple(sig.shape) == (n, ns) and tuple(col.shape) == (n, ns, 64) and tuple(skyb.shape) == (n, 64) and nosky.dtype == torch.uint8
    recomposed = (w[:, :, None] * (col.clamp(-1, 1) + 1)).sum(dim=1) + (1 - w.sum(dim=1, keepdim=True)) * (skyb.clamp(-1, 1) + 1) - 1
    assert float((recomposed - host).abs().max()) < 1e-5
    savg = sky_avg.reshape(1, 64)
    assert torch.equal(skyb[nosky.bool()], savg.expand(int(nosky.sum()), 64)) and torch.equal(skyb[~nosky.bool()], sky_c[~nosky.bool()])
    assert bool(nosky[vid[:, -1] != 0].all())


def _generator(weights_full, scene256, fast, aux=False):
    from oracle import ref_harness as RH
    RH.install("hip-fast" if fast else "hip")
    G, _ = RH.build_generator(weights_full, scene256)
    G = G.cuda()
    for p in G.parameters():
        p.requires_grad_(False)
    G.voxel.voxel_t = scene256.voxel_t.cuda()
    G.voxel.current_height_map = scene256.current_height_map.cuda()
    G.voxel.current_semantic_map = scene256.current_semantic_map.cuda()
    if fast:
        from scenedreamer_amd import dropin
        dropin.binding(G, aux=aux)
    return G, RH


@pytest.mark.needs_reference
@pytest.mark.parametrize("aux", [False, True])
def test_unmodified_generator_methods_on_the_fused_kernels(scene256, weights_full, aux):
    """Generator._forward_perpix / _forward_global of the UNMODIFIED reference with install_shims(fast=True): goldens recorded
    from the reference itself, and -- aux -- ALL twelve return values against the reference's own method on the same inputs."""
    import sys
    G, RH = _generator(weights_full, scene256, fast=True, aux=aux)
    from scenedreamer_amd import dropin
    voxlib = sys.modules["voxlib"]
    b = dropin.binding(G)
    for tag in "abc":
        g = golden(f"field_{tag}.npz")
        hw, ns = [int(v) for v in g["resolution_hw"]], int(g["num_samples"])
        RH.set_inference_overrides(G, ns, hw)
        z, ge = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(g["global_enc"]).cuda()
        cam = torch.from_numpy(g["cam_ori"])[None].cuda()
        with torch.no_grad():
            vid, d2, rd = voxlib.ray_voxel_intersection_perspective(
                G.voxel.voxel_t, torch.from_numpy(g["cam_ori"]), torch.from_numpy(g["cam_dir"]),
                torch.from_numpy(g["cam_up"]), float(g["cam_f"]), [float(v) for v in g["cam_c"]], G.cam_res, 6)
            vid, d2, rd = vid.unsqueeze(0), d2.unsqueeze(0), rd.unsqueeze(0)
            sky_in = voxlib.positional_encoding(rd.expand(-1, -1, -1, 1, -1).contiguous(), G.pe_params_sky[0], -1, G.pe_params_sky[1])
            G.sky_avg = torch.mean(G.sky_net(sky_in, z), dim=[1, 2], keepdim=True)
            before = dict(b.stats)
            out = G._forward_perpix(None, vid, d2.clone() if aux else d2, rd, cam, z, ge)
            img, raw = G._forward_global(out[0], z)
            ref = G._forward_perpix_reference(None, vid, d2.clone(), rd, cam, z, ge) if aux else None
        del G.sky_avg
        assert b.stats["perpix_fast"] == before["perpix_fast"] + 1 and b.stats["global_fast"] == before["global_fast"] + 1, b.stats
        assert len(out) == 12
        np.testing.assert_allclose(out[0].cpu().numpy(), g["net_out"], rtol=0, atol=TOL)
        np.testing.assert_allclose(img.cpu().numpy(), g["image"], rtol=0, atol=TOL)
        np.testing.assert_allclose(torch.tanh(raw).cpu().numpy(), img.cpu().numpy(), rtol=0, atol=1e-6)
        if not aux:
            assert all(o is None for o in out[1:])
            assert b.stats["tiles_in_place"] == before["tiles_in_place"] + 1       # d2 is the frame array itself: read in place
            continue
        names = dropin.PERPIX_OUTPUTS
        # new_dists, rand_depth: the reference's torch ops on the GPU accumulate the box lengths in float32, the kernel like
        # the CPU reference in double (mc_utils.py:102 on a CPU tensor): an ulp of a depth of ~100 voxels is 8e-6
        np.testing.assert_allclose(out[1].cpu().numpy(), ref[1].cpu().numpy(), rtol=0, atol=1e-6, err_msg=names[1])
        np.testing.assert_allclose(out[4].cpu().numpy(), ref[4].cpu().numpy(), rtol=0, atol=1e-4, err_msg=names[4])
        for i in (9, 10):                                  # sky_mask, sky_only_mask: exact
            assert torch.equal(out[i], ref[i]), names[i]
        assert out[11].dtype == ref[11].dtype == torch.int64 and float((out[11] != ref[11]).float().mean()) < 1e-3, names[11]
        np.testing.assert_allclose(out[2].cpu().numpy(), ref[2].cpu().numpy(), rtol=0, atol=TOL, err_msg="weights")
        np.testing.assert_allclose(out[3].cpu().numpy(), ref[3].cpu().numpy(), rtol=0, atol=TOL, err_msg="total_weights_raw")
        # net_out_s / net_out_c, per sample, rays that hit nothing included (the reference evaluates them at the camera origin).
        # sigma is the quantity the density head amplifies (a 1e-4 feature change moves it by ~1e-2): judged relative to its size
        assert all(o is not None for o in out) and [tuple(o.shape) for o in out] == [tuple(r.shape) for r in ref]
        for i, q99, worst in ((5, 5e-3, 0.25), (6, 2e-3, 0.05)):
            e = ((out[i] - ref[i]).abs() / (1 + ref[i].abs())).flatten()
            k = max(1, int(0.99 * e.numel()))
            assert float(e.kthvalue(k).values) < q99 and float(e.max()) < worst, (names[i], float(e.kthvalue(k).values), float(e.max()))
        np.testing.assert_allclose(out[7].cpu().numpy(), ref[7].cpu().numpy(), rtol=0, atol=TOL, err_msg="skynet_out_c")
        assert out[8].dtype == ref[8].dtype and float((out[8] != ref[8]).float().mean()) < 1e-3, "nosky_mask"
        print(f"golden {tag}: weights max abs diff vs the reference's own method {float((out[2] - ref[2]).abs().max()):.2e}")


@pytest.mark.needs_reference
def test_unsupported_calls_fall_to_the_reference_method_on_hip_ops(scene256, weights_full):
    """A call the fused kernel does not implement (box-boundary sampling) is served by the reference's own _forward_perpix --
    on the native modules -- and counted with its reason."""
    import sys
    G, RH = _generator(weights_full, scene256, fast=True)
    from scenedreamer_amd import dropin
    voxlib = sys.modules["voxlib"]
    g = golden("field_a.npz")
    hw, ns = [int(v) for v in g["resolution_hw"]], int(g["num_samples"])
    RH.set_inference_overrides(G, ns, hw)
    z, ge = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(g["global_enc"]).cuda()
    with torch.no_grad():
        vid, d2, rd = voxlib.ray_voxel_intersection_perspective(
            G.voxel.voxel_t, torch.from_numpy(g["cam_ori"]), torch.from_numpy(g["cam_dir"]), torch.from_numpy(g["cam_up"]),
            float(g["cam_f"]), [float(v) for v in g["cam_c"]], G.cam_res, 6)
        vid, d2, rd = vid.unsqueeze(0), d2.unsqueeze(0), rd.unsqueeze(0)
        fast = G._forward_perpix(None, vid, d2.clone(), rd, torch.from_numpy(g["cam_ori"])[None].cuda(), z, ge)[0]
        # the reference's method itself, module by module on the MFMA kernels, gives the same net_out
        slow = G._forward_perpix_reference(None, vid, d2.clone(), rd, torch.from_numpy(g["cam_ori"])[None].cuda(), z, ge)[0]
        assert G.render_net.__dict__.get("_sdn_backend") is not None          # LightningMLP ran natively inside it
        G.sample_use_box_boundaries = True
        G.num_samples = ns + 6
        b = dropin.binding(G)
        n0 = b.stats["perpix_reference"]
        out = G._forward_perpix(None, vid, d2.clone(), rd, torch.from_numpy(g["cam_ori"])[None].cuda(), z, ge)
    assert b.stats["perpix_reference"] == n0 + 1 and b.stats["why"].get("sample_use_box_boundaries") == 1
    assert out[0].shape == fast.shape and all(o is not None for o in out)
    np.testing.assert_allclose(slow.cpu().numpy(), g["net_out"], rtol=0, atol=TOL)
    np.testing.assert_allclose(fast.cpu().numpy(), slow.cpu().numpy(), rtol=0, atol=TOL)


@pytest.mark.needs_reference
@pytest.mark.parametrize("tile,coalesce", [(64, True), (64, False), (1024, True)])
def test_unmodified_inference_loop_on_the_fused_kernels(scene256, weights_full, tmp_path, tile, coalesce):
    """Generator.inference_givenstyle (scenedreamer.py:479-632), UNCHANGED, with install_shims(fast=True): 2 x 2 tiles (tile_size
    64) -- the frame evaluated ONCE when its first tile arrives and the tiles served as views of it (coalesce), or every tile
    evaluated in place from the frame arrays with the sky features of the pre-pass reused -- and one tile per frame (tile_size >=
    frame); against this package's renderer on the same trajectory: uint8 frames agree to one level."""
    from loop_helpers import run_reference_loop
    from scenedreamer_amd import camera, dropin, synth
    from scenedreamer_amd.output import to_uint8_hwc
    from scenedreamer_amd.renderer import Renderer
    G, _ = _generator(weights_full, scene256, fast=True)
    hw, ns, steps = [72, 104], 12, 3
    b = dropin.binding(G)
    b.coalesce = coalesce
    frames = run_reference_loop(G, str(tmp_path / "ref"), hw, ns, steps, tile_size=tile)
    tiles = 4 if tile == 64 else 1
    assert b.stats["perpix_fast"] == steps * tiles and b.stats["perpix_reference"] == 0 and b.stats["global_fast"] == steps * tiles, b.stats
    assert b.stats["tiles_in_place"] == steps * tiles and b.stats["sky_reused"] == steps * tiles, b.stats
    if tile == 64 and coalesce:
        assert b.stats["frames_coalesced"] == steps and b.stats["tiles_from_frame"] == b.stats["cnn_tiles_from_frame"] == steps * tiles, b.stats
    else:
        assert b.stats["frames_coalesced"] == b.stats["tiles_from_frame"] == b.stats["cnn_tiles_from_frame"] == 0, b.stats
    R = Renderer(weights_full, scene256, "cuda")
    R.set_style(synth.make_style(8888))
    poses = camera.eval_camera_poses(scene256, maxstep=steps, pattern=0, cam_ang=72)
    worst, off = 0, 0.0
    for f, img in enumerate(R.render_frames(poses, tuple(hw), ns, mode="fused")):
        mine = to_uint8_hwc(img).cpu().numpy().astype(np.int32)
        d = np.abs(frames[f].astype(np.int32) - mine)
        assert frames[f].shape == mine.shape == (hw[0], hw[1], 3) and frames[f].std() > 5
        worst, off = max(worst, int(d.max())), max(off, float((d > 0).mean()))
    print(f"inference_givenstyle (unmodified, fast shims, tile_size {tile}, coalesce {coalesce}) vs scenedreamer_amd frames: max |diff| {worst} level, "
          f"{100 * off:.2f} % of the values differ; stats {b.stats}")
    assert worst <= 1 and off < 0.2
    assert os.path.exists(os.path.join(str(tmp_path / "ref"), "rgb_render", "00002.png"))


@pytest.mark.needs_reference
def test_unmodified_loop_at_the_headline_config_against_oracle_tiles(weights_full, lut, tmp_path):
    """BASELINE config 2 (960x540, 24 samples/ray, scene 2048) through the reference's OWN loop, default tile_size 128: the binding
    evaluates the padded frame once when the first of its 40 tiles arrives and serves the tiles as views (dropin.frame_field /
    frame_image_tile).  The loop's float pixels (the frame the views are cut from, before the loop's uint8 conversion) against the
    CPU oracle on 5 tiles of the reference's tile grid -- two frame corners incl. the ragged edge tile, the centre, two more."""
    from loop_helpers import run_reference_loop
    from oracle import field_ref as FR
    from scenedreamer_amd import camera, dropin, synth
    scene = synth.make_scene(2048, 3407)
    G, _ = _generator(weights_full, scene, fast=True)
    hw, ns, steps = [540, 960], 24, 3
    b = dropin.binding(G)
    got, cams = [], []
    b.on_frame = lambda fr: got.append((fr["img"].clone(), fr["net_out"].shape))
    # the camera the LOOP uses: EvalCameraController runs on the generator's own voxel handle, whose trans_mat is a registered buffer
    # and moved to the GPU with the generator (camctl.py:9-60) -- its matrix products there round differently from the host's
    # (cam_ori 32.999996 instead of 33.0 on this scene), and the field is chaotic in such an ulp at a few dozen rays per frame
    # (tools/dbg_dropin_err.py: 22 rays > 1e-3 on net_out).  The oracle is given the very poses the loop's ray caster received.
    import sys
    voxlib = sys.modules["imaginaire.generators.scenedreamer"].voxlib      # (the package object the loop's module bound at import)
    rvip = voxlib.ray_voxel_intersection_perspective

    def recording_rvip(voxel_t, cam_ori, cam_dir, cam_up, cam_f, cam_c, img_dims, max_samples):
        cams.append((torch.as_tensor(cam_ori).detach().cpu().numpy().astype(np.float32), torch.as_tensor(cam_dir).detach().cpu().numpy().astype(np.float32),
                     torch.as_tensor(cam_up).detach().cpu().numpy().astype(np.float32), float(cam_f)))
        return rvip(voxel_t, cam_ori, cam_dir, cam_up, cam_f, cam_c, img_dims, max_samples)
    voxlib.ray_voxel_intersection_perspective = recording_rvip
    try:
        frames = run_reference_loop(G, str(tmp_path / "ref"), hw, ns, steps, tile_size=128)
    finally:
        b.on_frame = None
        voxlib.ray_voxel_intersection_perspective = rvip
    assert len(got) == steps and b.stats["frames_coalesced"] == steps and b.stats["tiles_from_frame"] == 40 * steps, b.stats
    assert b.stats["perpix_reference"] == 0 and b.stats["global_reference"] == 0, b.stats
    img, no_shape = got[-1]
    assert tuple(no_shape) == (1, 570, 990, 64) and tuple(img.shape) == (1, 3, 570, 990)
    img = img[:, :, 15:-15, 15:-15].cpu().numpy()
    # the uint8 frame the loop handed to its writer is this float image
    u8 = np.clip(np.floor((np.transpose(img[0], (1, 2, 0)) * 0.5 + 0.5) * 255), 0, 255).astype(np.int32)    # write_img truncates (scenedreamer.py:513)
    assert np.abs(u8 - frames[-1].astype(np.int32)).max() <= 1
    pose = camera.eval_camera_poses(scene, maxstep=steps, pattern=0, cam_ang=72)[steps - 1]
    assert len(cams) == steps and cams[-1][3] == pose[3] * (hw[1] - 1)                      # the intrinsics are host arithmetic: equal
    assert max(float(np.abs(cams[-1][k] - pose[k].numpy()).max()) for k in range(3)) < 1e-5   # the extrinsics: equal to an ulp or so
    pose = (torch.from_numpy(cams[-1][0]), torch.from_numpy(cams[-1][1]), torch.from_numpy(cams[-1][2]), pose[3])
    R_z = G.style_net(torch.from_numpy(np.asarray(synth.make_style(8888))).cuda())
    from scenedreamer_amd.renderer import Renderer
    R = Renderer(weights_full, scene, "cuda")
    R.set_style(synth.make_style(8888))
    assert float((R.z - R_z).abs().max()) < 1e-5
    tiles = [(0, 0), (4, 7), (2, 3), (1, 6), (3, 1)]
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    ref = FR.render_frame_tiled(weights_full, lut, scene.voxel_t.numpy(), (pose[0].numpy(), pose[1].numpy(), pose[2].numpy(), pose[3]),
                                tuple(hw), ns, R.z.cpu().numpy(), R.global_enc.cpu().numpy(), tiles=tiles)
    worst = 0.0
    for t in tiles:
        r0, c0, tile = ref[t]
        tile = tile.numpy()
        e = float(np.abs(img[:, :, r0:r0 + tile.shape[2], c0:c0 + tile.shape[3]] - tile).max())
        worst = max(worst, e)
        assert np.isfinite(tile).all(), t
    print(f"unmodified inference_givenstyle, 960x540x24, scene 2048, tile_size 128 (frame evaluated once): max abs err vs oracle on {len(tiles)} tiles {worst:.3e}")
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/dropin_loop_error.json", "w") as f:
        json.dump({"config": "960x540x24, scene 2048, unmodified inference_givenstyle on install_shims(fast=True), tile_size 128, frame evaluated once "
                             "for its 40 tiles; last of 3 frames", "tiles": [list(t) for t in tiles], "max_abs_err": worst}, f)
    assert worst < TOL
