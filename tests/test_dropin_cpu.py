"""Host logic of the drop-in surface (scenedreamer_amd/modules.py, dropin.py) -- no GPU: the import hook on the unmodified
reference package, state-dict compatibility of the stand-alone module classes, the composite forwards, the tile -> frame
window arithmetic and the staleness tracking of the backends."""
import sys

import numpy as np
import pytest
import torch


def _tile_views(H0, W0, M, tile, pad):
    """The frame-wide arrays and the tile views exactly as inference_givenstyle cuts them (scenedreamer.py:576-616)."""
    vid = torch.arange(H0 * W0 * M, dtype=torch.int32).reshape(H0, W0, M, 1)
    d2 = torch.arange(2 * H0 * W0 * M, dtype=torch.float32).reshape(2, H0, W0, M, 1)
    rd = torch.arange(H0 * W0 * 3, dtype=torch.float32).reshape(H0, W0, 1, 3)
    vid_all, d2_all, rd_all = vid.unsqueeze(0), d2.unsqueeze(0), rd.unsqueeze(0)
    nh, nw = (H0 - pad + tile - 1) // tile, (W0 - pad + tile - 1) // tile
    for ih in range(nh):
        for iw in range(nw):
            hb, he = ih * tile, min(ih * tile + tile + pad, H0)
            wb, we = iw * tile, min(iw * tile + tile + pad, W0)
            yield (vid_all[:, hb:he, wb:we, :, :], d2_all[:, :, hb:he, wb:we, :, :], rd_all[:, hb:he, wb:we, :, :],
                   (vid, d2, rd), (hb, he, wb, we))


@pytest.mark.parametrize("H0,W0,M,tile,pad", [(23, 31, 6, 8, 6), (20, 17, 6, 64, 4), (9, 40, 1, 8, 2), (12, 12, 3, 4, 0), (7, 1, 6, 3, 0)])
def test_frame_window_addresses_the_tile_inside_the_frame_arrays(H0, W0, M, tile, pad):
    from scenedreamer_amd.dropin import GeneratorBinding
    n = 0
    for v, d, r, (vid, d2, rd), (hb, he, wb, we) in _tile_views(H0, W0, M, tile, pad):
        fw = GeneratorBinding.frame_window(v, d, r)
        assert fw is not None, (hb, he, wb, we)
        win, (pv, pd, pr) = fw
        assert (pv, pd, pr) == (vid.data_ptr(), d2.data_ptr(), rd.data_ptr()) and win.n_src == H0 * W0
        h, w = he - hb, we - wb
        assert win.n_rays == h * w
        # RayWindow::src (csrc/field.hip): local ray -> source ray
        loc = np.arange(h * w)
        src = win.first + (loc // win.cols) * win.pitch + (loc % win.cols)
        np.testing.assert_array_equal(vid.reshape(H0 * W0, M).numpy()[src], v.reshape(h * w, M).numpy())
        np.testing.assert_array_equal(d2.reshape(2, H0 * W0, M).numpy()[:, src], d.reshape(2, h * w, M).numpy())
        np.testing.assert_array_equal(rd.reshape(H0 * W0, 3).numpy()[src], r.reshape(h * w, 3).numpy())
        n += 1
    assert n >= 1


def test_frame_window_refuses_what_is_not_one_tile_of_three_frames():
    from scenedreamer_amd.dropin import GeneratorBinding
    v, d, r, _, _ = next(_tile_views(20, 24, 6, 8, 4))
    assert GeneratorBinding.frame_window(v, d, r) is not None
    assert GeneratorBinding.frame_window(v.contiguous(), d, r) is None                 # a copy of the tile: another storage offset
    assert GeneratorBinding.frame_window(v, d, r[:, :, 1:]) is None                    # shapes disagree
    v2, d2, r2, _, _ = list(_tile_views(20, 24, 6, 8, 4))[1]
    assert GeneratorBinding.frame_window(v, d2, r) is None                             # tiles disagree
    fw = GeneratorBinding.frame_window(v[:, ::2], d[:, :, ::2], r[:, ::2])              # every second row IS a window (pitch 2 W)
    assert fw is not None and fw[0].pitch == 2 * 24 and fw[0].n_rays == v[:, ::2].shape[1] * v.shape[2]
    assert GeneratorBinding.frame_window(v[:, :, ::2], d[:, :, :, ::2], r[:, :, ::2]) is None  # strided columns are not a window


def test_backend_tracks_live_parameters():
    from scenedreamer_amd import modules
    m = modules.SKYMLP(33, 256, 64)
    B = modules.Backend()
    assert B.bind("sky_net.", m) is True and B.bind("sky_net.", m) is False
    assert B.w["sky_net.fc1.weight"].data_ptr() == m.fc1.weight.data_ptr()              # an alias, not a copy
    with torch.no_grad():
        m.fc3.bias.add_(1.0)                                                             # in-place update (optimizer step / load_state_dict)
    assert B.bind("sky_net.", m) is True and B.bind("sky_net.", m) is False
    m.load_state_dict(modules.SKYMLP(33, 256, 64).state_dict())
    assert B.bind("sky_net.", m) is True
    z = torch.randn(1, 256)
    calls = []
    fold = lambda b, zz: calls.append(zz.clone())
    B.style("sky_net.", z, 0, fold)
    B.style("sky_net.", z, 0, fold)
    assert len(calls) == 1
    z.mul_(2.0)
    B.style("sky_net.", z, 0, fold)
    assert len(calls) == 2 and torch.equal(calls[1], z)


def test_backend_style_is_not_fooled_by_a_recycled_address():
    """The reference builds z = style_net(style) afresh per call and frees the old one: the allocator hands the next z the same
    address with version 0 again.  The backend keeps the keyed tensor alive, so "same address" can only mean "same storage"."""
    import gc
    from scenedreamer_amd import modules
    B = modules.Backend()
    calls = []
    fold = lambda b, zz: calls.append(zz.clone())
    z1 = torch.full((1, 256), 1.0)
    p1 = z1.data_ptr()
    B.style("render_net.", z1, 0, fold)
    assert B._zref["render_net."] is z1
    del z1
    gc.collect()
    z2 = torch.full((1, 256), 2.0)                 # would land on z1's address if z1 had been released
    assert z2.data_ptr() != p1                     # ... it was not: the backend holds it
    B.style("render_net.", z2, 0, fold)
    assert len(calls) == 2 and float(calls[1][0, 0]) == 2.0
    v = z2[0:1]                                    # a view of the folded tensor: same content, no refold
    B.style("render_net.", v, 0, fold)
    assert len(calls) == 2
    # a rebind (new weights) forgets the style: the next call folds again
    m = modules.SKYMLP(33, 256, 64)
    B.bind("render_net.", m)
    B.style("render_net.", z2, 0, fold)
    assert len(calls) == 3


def test_one_row_tiles_only_coalesce_when_the_frame_pitch_is_known(monkeypatch):
    """frame_window cannot see the frame's row pitch in a one-row tile: rays() recovers it from the sky pre-pass's
    ray directions or marks the window as not coalescible."""
    from scenedreamer_amd import dropin, fused, modules
    H0, W0, M = 7, 10, 6
    vid = torch.zeros((1, H0, W0, M, 1), dtype=torch.int32)
    d2 = torch.zeros((1, 2, H0, W0, M, 1))
    rd = torch.zeros((1, H0, W0, 1, 3))
    v, d, r = vid[:, 6:7, 4:9], d2[:, :, 6:7, 4:9], rd[:, 6:7, 4:9]
    win, bases = dropin.GeneratorBinding.frame_window(v, d, r)
    assert win.pitch == 5 and win.first == 6 * W0 + 4 and win.n_rays == 5       # pitch is a guess here

    class Sky:
        pass
    b = dropin.GeneratorBinding()
    b.B._zkey["sky_net."] = ("k",)
    G = type("G", (), {})()
    G.sky_net = Sky()
    monkeypatch.setattr(modules.Backend, "tensors_key", staticmethod(lambda m: "w"))
    Sky._sdn_native = True
    for ref_shape, known in (((1, H0, W0, 1, 3), True), ((H0 * W0, 3), False)):
        G.sky_net.__dict__["_sdn_last_frame"] = dict(rd_ref=rd.view(ref_shape), rd_ptr=rd.data_ptr(), rd_version=rd._version,
                                                     n_rays=H0 * W0, sky_c="sky", zkey=("k",), wkey="w")
        win, *_ = b.rays(G, v, d, r)
        assert getattr(win, "pitch_known") is known and win.pitch == (W0 if known else 5)


def test_composite_forwards_without_a_gpu():
    """CPU tensors (or autograd) go through the composite forward and say why."""
    from scenedreamer_amd import modules
    torch.manual_seed(0)
    net = modules.LightningMLP(128, 256, 0, mask_dim=12, out_channels_s=1, out_channels_c=64)
    x, z = torch.randn(1, 2, 3, 4, 128), torch.randn(1, 256)
    m = torch.zeros(1, 2, 3, 4, 12)
    m[..., 5] = 1
    s, c = net(x, None, z, m)
    assert tuple(s.shape) == (1, 2, 3, 4, 1) and tuple(c.shape) == (1, 2, 3, 4, 64) and s.requires_grad
    assert "CUDA" in net._sdn_composite_reason
    # ModLinear with N = 1 is the plain layer the kernels fold it into (renderer.fold_render_net)
    from scenedreamer_amd.renderer import fold_render_net
    R = modules.Backend()
    R.bind("render_net.", net)
    fold_render_net(R, z)
    f = torch.nn.functional.leaky_relu(net.fc_1(x) + net.fc_m_a(m), 0.2)
    want = net.fc_2(f, z[:, None, None, None, :])
    got = torch.nn.functional.linear(f, R.mod[2][0], R.mod[2][1])
    np.testing.assert_allclose(got.detach().numpy(), want.detach().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose((net.fc_m_a(m) + net.fc_1.bias)[0, 0, 0, 0].detach().numpy(), R.label_bias[5].numpy(), rtol=0, atol=1e-6)


@pytest.mark.needs_reference
def test_import_hook_patches_the_unmodified_reference_package():
    from oracle import ref_harness as RH
    from scenedreamer_amd import dropin, modules, synth
    try:
        RH.install("hip-fast")
        G, _ = RH.build_generator(None, synth.make_scene(64, 3407))
        import imaginaire.generators.gancraft_base as gb
        import imaginaire.generators.scenedreamer as sd
        import imaginaire.model_utils.layers as ly
        assert sd.LightningMLP is ly.LightningMLP and isinstance(G.render_net, ly.LightningMLP)
        assert modules.is_native(G.render_net) and modules.is_native(G.sky_net) and modules.is_native(G.denoiser)
        assert type(G.sky_net) is gb.SKYMLP and type(G.denoiser) is gb.RenderCNN
        assert G._forward_perpix.__func__ is dropin.fast_forward_perpix and G._forward_global.__func__ is dropin.fast_forward_global
        assert callable(G._forward_perpix_reference) and callable(G._forward_global_reference)
        # the reference's files were executed unchanged: its classes' own forwards are the composite paths
        assert ly.LightningMLP._forward_composite.__code__.co_filename.endswith("imaginaire/model_utils/layers.py")
        # the stand-alone classes carry the same parameters (names and shapes) as the reference's
        for mine, ref in ((modules.LightningMLP(128, 256, 0, mask_dim=12, out_channels_s=1, out_channels_c=64), G.render_net),
                          (modules.SKYMLP(33, 256, 64), G.sky_net), (modules.RenderCNN(64, 256), G.denoiser)):
            assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
            mine.load_state_dict(ref.state_dict())
        # ... and compute the same function
        torch.manual_seed(1)
        x, z = torch.randn(1, 2, 3, 4, 128), torch.randn(1, 256)
        m = torch.zeros(1, 2, 3, 4, 12)
        m[..., 7] = 1
        mine = modules.LightningMLP(128, 256, 0, mask_dim=12, out_channels_s=1, out_channels_c=64)
        mine.load_state_dict(G.render_net.state_dict())
        with torch.no_grad():
            for a, b in zip(mine(x, None, z, m), G.render_net(x, None, z, m)):
                np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=1e-5)
            sk = modules.SKYMLP(33, 256, 64)
            sk.load_state_dict(G.sky_net.state_dict())
            pe = torch.randn(1, 5, 6, 1, 33)
            np.testing.assert_allclose(sk(pe, z).numpy(), G.sky_net(pe, z).numpy(), rtol=0, atol=1e-5)
            cn = modules.RenderCNN(64, 256)
            cn.load_state_dict(G.denoiser.state_dict())
            xx = torch.randn(1, 64, 7, 9)
            np.testing.assert_allclose(cn(xx, z).numpy(), G.denoiser(xx, z).numpy(), rtol=0, atol=1e-5)
        # a CPU call of the bound method is served by the reference's own method (counted, with the reason)
        b = dropin.binding(G)
        assert b.why_not_perpix(G, torch.zeros(1, 2, 2, 6, 1, dtype=torch.int32), torch.zeros(1, 2, 2, 2, 6, 1), torch.zeros(1, 2, 2, 1, 3),
                                torch.zeros(1, 3), z, torch.zeros(1, 2)) is not None
    finally:
        dropin.uninstall_import_hook()
        RH.install("oracle")      # leave the process as the other CPU tests expect it
        for name in [n for n in sys.modules if n.split(".")[0] == "imaginaire"]:
            del sys.modules[name]


def test_frame_image_tile_maps_tile_views_of_the_cached_frame():
    """GeneratorBinding.frame_image_tile: a tile-shaped VIEW of the frame's cached net_out is answered with the same window of
    the frame's image; a copy, a strided view or an unrelated tensor is not (the per-tile CNN path serves those)."""
    from scenedreamer_amd.dropin import GeneratorBinding
    b = GeneratorBinding()
    H0, W0 = 23, 31
    full = torch.arange(H0 * W0 * 64, dtype=torch.float32).reshape(1, H0, W0, 64)
    img = torch.arange(3 * H0 * W0, dtype=torch.float32).reshape(1, 3, H0, W0)
    b._frame = dict(net_out=full, img=img, raw=img + 0.5, img_key=None)
    z = torch.zeros(1, 4)

    class _G:
        denoiser = None
    b._frame["img_key"] = ((z.data_ptr(), z._version), None)      # the image of this style is already there: no CNN call
    b._frame["img_z"] = z                                          # (the keyed tensor is held, so its address stays its own)
    for hb, he, wb, we in ((0, 14, 0, 14), (8, 23, 16, 31), (3, 4, 5, 31), (0, 23, 30, 31)):
        tile = full[:, hb:he, wb:we, :]
        got = b.frame_image_tile(_G, tile, z)
        assert got is not None, (hb, he, wb, we)
        assert torch.equal(got[0], img[:, :, hb:he, wb:we]) and torch.equal(got[1], img[:, :, hb:he, wb:we] + 0.5)
    assert b.stats["cnn_tiles_from_frame"] == 4
    assert b.frame_image_tile(_G, full[:, 2:9, 3:9, :].clone(), z) is None              # a copy: another storage
    assert b.frame_image_tile(_G, full[:, ::2, :, :], z) is None                         # strided rows
    assert b.frame_image_tile(_G, full[:, :, :, :32], z) is None                         # not 64 features
    assert b.frame_image_tile(_G, torch.zeros(1, 4, 4, 64), z) is None
    b._frame = None
    assert b.frame_image_tile(_G, full[:, :4, :4, :], z) is None
