"""Building blocks of the fused field path on the MI355X: MFMA operand layouts, table collapse, encode."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def test_mfma_layout_probe():
    """A[32,16] x B[16,32] through v_mfma_f32_32x32x16_f16 with the lane/register maps field.hip assumes.
    Asymmetric integer data so that any row/column or k permutation error shows up."""
    from scenedreamer_amd import capi
    lib = capi.lib()
    rng = np.random.default_rng(0)
    A = rng.integers(-8, 9, size=(32, 16)).astype(np.float32)
    B = rng.integers(-8, 9, size=(16, 32)).astype(np.float32)
    A[3, :] = 0.0
    A[3, 5] = 2.0 ** -20      # f16 subnormal input must not be flushed (the lo parts of the split live there)
    B[5, 7] = 1024.0
    a, b = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    c = torch.zeros(32, 32, device="cuda")
    capi.check(lib.sdn_debug_mfma_probe(a.data_ptr(), b.data_ptr(), c.data_ptr(), capi.current_stream()))
    ref = A.astype(np.float64) @ B.astype(np.float64)
    np.testing.assert_allclose(c.cpu().numpy(), ref, rtol=0, atol=1e-6)


@pytest.fixture(scope="module")
def renderer(weights_full, scene256):
    from scenedreamer_amd import synth
    from scenedreamer_amd.renderer import Renderer
    r = Renderer(weights_full, scene256, "cuda")
    r.set_style(synth.make_style(8888))
    return r


def test_collapsed_table_equals_5d_lookup(renderer, oracle, weights_full):
    """Features from the per-scene collapsed 3-D table == the reference's 32-corner 5-D blend (oracle)."""
    from scenedreamer_amd import fused
    sc = fused.prepare_scene(renderer)
    T = sc["T"]
    rng = np.random.default_rng(1)
    B = 4096
    x3 = rng.random((B, 3), dtype=np.float32)
    genc01 = ((sc["genc"] + 1) / 2).astype(np.float32)
    x5 = np.concatenate([x3, np.broadcast_to(genc01, (B, 2))], axis=1).astype(np.float32)
    S = np.float32(renderer.grid_S)
    ref = oracle.grid_encode_fwd(x5, weights_full["hash_encoder.embeddings"], weights_full["hash_encoder.offsets"], S, 16)
    table = sc["table3"].cpu().numpy()          # [16, T, 8]
    scales = sc["scales"].cpu().numpy()
    P1, P2 = np.uint32(2654435761), np.uint32(805459861)
    got = np.zeros_like(ref)
    for l in range(16):
        pos = (x3 * scales[l]).astype(np.float32) + np.float32(0.5)
        pg = np.floor(pos).astype(np.uint32)
        fr = (pos - pg).astype(np.float32)
        acc = np.zeros((B, 8), np.float32)
        for c in range(8):
            cb = [(c >> d) & 1 for d in range(3)]
            w = np.ones(B, np.float32)
            for d in range(3):
                w = w * (fr[:, d] if cb[d] else (np.float32(1) - fr[:, d]))
            with np.errstate(over="ignore"):
                h = (pg[:, 0] + np.uint32(cb[0])) ^ ((pg[:, 1] + np.uint32(cb[1])) * P1) ^ ((pg[:, 2] + np.uint32(cb[2])) * P2)
            acc += w[:, None] * table[l][h & np.uint32(T - 1)]
        got[l] = acc
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_encode_matches_oracle(renderer, oracle, weights_full, lut, tag):
    """Sample placement (discrete decisions bit-exact) and features of encode_kernel vs the CPU oracle."""
    from oracle import field_ref as FR
    from scenedreamer_amd import fused
    g = golden(f"field_{tag}.npz")
    renderer.set_style_code(g["z"])
    renderer.global_enc = torch.from_numpy(g["global_enc"]).cuda()
    renderer._fused_scene = None
    M = g["voxel_id"].shape[-2]
    ns = int(g["num_samples"])
    vid = torch.from_numpy(g["voxel_id"]).cuda().reshape(-1, M)
    d2 = torch.from_numpy(g["depth2"]).cuda().reshape(2, -1, M)
    rd = torch.from_numpy(g["raydirs"]).cuda().reshape(-1, 3)
    R = vid.shape[0]
    buf = fused.encode(renderer, vid, d2, rd, torch.from_numpy(g["cam_ori"]), ns)
    torch.cuda.synchronize()
    _, aux = FR.forward_perpix(weights_full, lut, renderer.voxel_t.shape, g["voxel_id"], g["depth2"], g["raydirs"],
                               g["cam_ori"][None], g["z"], g["global_enc"], ns, sky_avg=g["sky_avg"], return_aux=True)
    nch = (ns + 3) // 4
    ntile = (R + 7) // 8
    # the features are stored as the MLP's operands: per lane and k-step 8 f16 hi + 8 f16 lo (hi + lo = the value to 2^-22)
    fh = buf["feat"].cpu().numpy().view(np.float16).reshape(ntile, nch, 8, 64, 2, 8).astype(np.float32)
    feat = fh[..., 0, :] + fh[..., 1, :]
    dist = buf["dist"].cpu().numpy().reshape(ntile, nch, 32)
    label = buf["label"].cpu().numpy().reshape(ntile, nch, 32)
    # un-permute: lane = h*32 + 4*ray_in_tile + sample_in_step ; level = 2*s + h
    ref_feat = aux["feature_in"].numpy().reshape(R, ns, 16, 8)
    ref_dist = (aux["new_dists"].numpy().reshape(R, ns) * np.float32(0.25)).astype(np.float32)
    ref_idx = aux["new_idx"].numpy().reshape(R, ns)
    lutt = np.asarray(lut)
    red = lutt[g["voxel_id"].reshape(R, M)]
    red[red == 0] = 3
    ref_label = np.take_along_axis(red, ref_idx, axis=1)
    got_feat = np.zeros_like(ref_feat)
    got_dist = np.zeros_like(ref_dist)
    got_label = np.zeros_like(ref_label)
    for ray in range(R):
        t, rt = divmod(ray, 8)
        for smp in range(ns):
            ch, si = divmod(smp, 4)
            j = 4 * rt + si
            got_dist[ray, smp] = dist[t, ch, j]
            got_label[ray, smp] = label[t, ch, j]
            for h in range(2):
                got_feat[ray, smp, h::2] = feat[t, ch, :, h * 32 + j, :]
    np.testing.assert_array_equal(got_label, ref_label)
    np.testing.assert_array_equal(got_dist.view(np.int32), ref_dist.view(np.int32))
    # rays that hit nothing get weight 0 (scenedreamer.py:376); the kernel does not gather / write their features
    sky_only = g["voxel_id"].reshape(R, M)[:, 0] == 0
    assert (~sky_only).sum() > 0
    np.testing.assert_allclose(got_feat[~sky_only], ref_feat[~sky_only], rtol=0, atol=5e-6)
    # ray flags
    flags = buf["rayflag"].cpu().numpy()
    np.testing.assert_array_equal((flags & 1).astype(bool), sky_only)
    wc = aux["worldcoord2"].numpy().reshape(R, ns, 3)
    nosky = (g["voxel_id"].reshape(R, M)[:, -1] != 0) | (wc[:, :, 0] <= 1.0).any(axis=1)
    np.testing.assert_array_equal(((flags >> 1) & 1).astype(bool), nosky)


def test_sky_mlp_kernel_matches_torch(renderer):
    """sky_kernel (PE + 6-layer MLP on MFMA + frame mean) vs posenc kernel + PyTorch fp32 GEMMs."""
    from scenedreamer_amd import fused
    g = golden("field_a.npz")
    renderer.set_style_code(g["z"])
    rd = torch.from_numpy(g["raydirs"]).cuda().reshape(-1, 3)
    rd = torch.cat([rd, rd.flip(0)[:77]])                       # ragged count (not a multiple of 32 / 128)
    ref = renderer.sky_features(rd)
    got, avg = fused.sky_fused(renderer, rd)
    assert (got - ref).abs().max().item() < 2e-4
    assert (avg - ref.mean(dim=0, keepdim=True)).abs().max().item() < 2e-5
    # opt-in: hidden layers fc2..fc5 as f16 Whi.Xhi + block-scaled fp6 corrections (the field MLP's colour-layer scheme)
    try:
        renderer.sky_terms = 6
        got6, avg6 = fused.sky_fused(renderer, rd)
    finally:
        renderer.sky_terms = None
    assert (got6 - ref).abs().max().item() < 2e-4 and (avg6 - ref.mean(dim=0, keepdim=True)).abs().max().item() < 2e-5


def test_sample_depth_op_matches_reference_golden():
    """ops.sample_depth_batched (sdn_sample_depth) with the reference's signature: deterministic and stochastic
    (training: stratified random, mc_utils.py:121-125) placement against vectors recorded from the unmodified
    mc_utils.sample_depth_batched.  Box indices, depths and distances bit-exact (NaN positions included)."""
    from conftest import bits
    from scenedreamer_amd import ops
    g = golden("stochastic_sampling.npz")
    d2 = torch.from_numpy(g["depth2"]).cuda()
    for ns in (13, 25):
        # (the goldens were recorded from the reference's CPU run: rand / nsamples as an IEEE division)
        rd, nd, idx = ops.sample_depth_batched(d2, ns, deterministic=False, use_box_boundaries=False, sample_depth=3,
                                               rand=torch.from_numpy(g[f"u{ns}"]).cuda(), division="ieee")
        assert rd.shape == g[f"depth{ns}"].shape and idx.dtype == torch.int64
        np.testing.assert_array_equal(idx.cpu().numpy().astype(np.int8), g[f"idx{ns}"])
        np.testing.assert_array_equal(bits(rd.cpu().numpy()), bits(g[f"depth{ns}"]))
        np.testing.assert_array_equal(bits(nd.cpu().numpy()), bits(g[f"dists{ns}"]))
        rd, nd, idx = ops.sample_depth_batched(d2, ns, deterministic=True, use_box_boundaries=False, sample_depth=3)
        np.testing.assert_array_equal(idx.cpu().numpy().astype(np.int8), g[f"det_idx{ns}"])
        np.testing.assert_array_equal(bits(rd.cpu().numpy()), bits(g[f"det_depth{ns}"]))
        np.testing.assert_array_equal(bits(nd.cpu().numpy()), bits(g[f"det_dists{ns}"]))
    # seeded default draw == the reference's own draw (same generator call, same shape)
    torch.manual_seed(7)
    a = ops.sample_depth_batched(d2, 13, deterministic=False, use_box_boundaries=False, sample_depth=3)
    torch.manual_seed(7)
    u = torch.rand([1, d2.shape[2], d2.shape[3], 13, 1], dtype=torch.float32, device="cuda")
    b = ops.sample_depth_batched(d2, 13, deterministic=False, use_box_boundaries=False, sample_depth=3, rand=u)
    assert torch.equal(a[2], b[2]) and torch.equal(a[0].nan_to_num(-1), b[0].nan_to_num(-1))


def test_stratified_positions_follow_the_gpu_reference_division():
    """`rand_samples / nsamples` (mc_utils.py:123) on a CUDA tensor is a multiplication by the float32 reciprocal in PyTorch,
    on a CPU tensor an IEEE division.  The op's default (division="reciprocal") must reproduce what the reference's lines
    give when they run on THIS GPU; rays with a single box are used so that the cumulative box depth -- where the CUDA
    and CPU cumsum kernels differ too -- is the same number either way: new_dists then has to agree bit for bit."""
    from conftest import bits
    from scenedreamer_amd import ops
    torch.manual_seed(11)
    n, M = 4096, 6
    t = torch.full((1, 1, n, M, 1), float("nan"), device="cuda")
    t2 = t.clone()
    t[..., 0, :] = 1 + 4 * torch.rand(1, 1, n, 1, device="cuda")
    t2[..., 0, :] = t[..., 0, :] + 0.05 + 3.5 * torch.rand(1, 1, n, 1, device="cuda")
    d2 = torch.stack([t, t2], dim=1)
    differ = 0
    for ns in (13, 25, 41):
        u = torch.rand([1, 1, n, ns, 1], device="cuda")
        total = (t2[..., :1, :] - t[..., :1, :]).clamp(max=3.0)
        s = u / ns                                                              # the reference's expression, on the GPU
        s = (s + torch.linspace(0, 1, ns + 1, device="cuda")[:-1].view(1, 1, 1, ns, 1)) * total
        want = s[..., 1:, :] - s[..., :-1, :]
        got = ops.sample_depth_batched(d2, ns, deterministic=False, use_box_boundaries=False, sample_depth=3, rand=u)[1]
        np.testing.assert_array_equal(bits(got.cpu().numpy()), bits(want.cpu().numpy()))
        ieee = ops.sample_depth_batched(d2, ns, deterministic=False, use_box_boundaries=False, sample_depth=3, rand=u,
                                        division="ieee")[1]
        differ += int((ieee != got).sum())
    assert differ > 0       # the two conventions are distinguishable on this input, so the test above means something


@pytest.mark.needs_reference
def test_sample_depth_with_box_boundaries_matches_reference():
    """use_box_boundaries=True (the reference signature's default) against the UNMODIFIED mc_utils.sample_depth_batched run
    on the CPU with the same two random draws (torch.rand_like / torch.rand are handed the tensors)."""
    from unittest import mock

    from oracle import ref_harness as RH
    from scenedreamer_amd import ops
    RH.install("oracle")
    from imaginaire.model_utils.gancraft import mc_utils
    g = golden("stochastic_sampling.npz")
    d2 = torch.from_numpy(g["depth2"])
    N, _, H, W, M, _ = d2.shape
    torch.manual_seed(3)
    for ns, det in ((13, False), (25, False), (13, True)):
        ub, u = torch.rand(N, H, W, M, 1), torch.rand(N, H, W, ns, 1)
        with mock.patch.object(torch, "rand_like", lambda x: ub.clone()), mock.patch.object(torch, "rand", lambda *a, **k: u.clone()):
            ref = mc_utils.sample_depth_batched(d2.clone(), ns, deterministic=det, use_box_boundaries=True, sample_depth=3)
        got = ops.sample_depth_batched(d2.cuda(), ns, deterministic=det, use_box_boundaries=True, sample_depth=3,
                                       rand=u.cuda(), boundary_rand=ub.cuda(), division="ieee")
        assert got[0].shape == ref[0].shape == (N, H, W, ns + M, 1) and got[2].dtype == torch.int64
        same = (got[2].cpu() == ref[2])
        assert float(same.float().mean()) > 0.999          # (cumsum: float scan on the GPU, double accumulation on the CPU)
        np.testing.assert_allclose(got[1].cpu().numpy(), ref[1].numpy(), rtol=0, atol=2e-6, equal_nan=True)
        a, b = got[0].cpu()[same], ref[0][same]
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=2e-5, equal_nan=True)


def test_fused_encode_with_stochastic_sampling(renderer, weights_full, lut):
    """The fused field path with training-time stochastic sample placement (u = torch.rand draw) against the oracle
    evaluated with the same randoms: net_out within 1e-3."""
    from oracle import field_ref as FR
    from scenedreamer_amd import fused
    g = golden("field_b.npz")
    M = g["voxel_id"].shape[-2]
    hp, wp = g["net_out"].shape[1:3]
    ns = int(g["num_samples"])
    torch.manual_seed(5)
    u = torch.rand([1, hp, wp, ns + 1, 1], dtype=torch.float32)
    orig = FR.sample_depth_batched
    FR.sample_depth_batched = lambda d2, nsamples, sd: orig(d2, nsamples, sd, rand=u)
    try:
        ref = FR.forward_perpix(weights_full, lut, (int(renderer.voxel_dims[0]), int(renderer.voxel_dims[1]), int(renderer.voxel_dims[2])),
                                g["voxel_id"], g["depth2"], g["raydirs"], g["cam_ori"][None], g["z"], g["global_enc"], ns,
                                sky_avg=g["sky_avg"])
    finally:
        FR.sample_depth_batched = orig
    renderer.set_style_code(g["z"])
    renderer.global_enc = torch.from_numpy(g["global_enc"]).cuda()
    vid = torch.from_numpy(g["voxel_id"]).cuda().reshape(-1, M)
    d2 = torch.from_numpy(g["depth2"]).cuda().reshape(2, -1, M)
    rd = torch.from_numpy(g["raydirs"]).cuda().reshape(-1, 3)
    with torch.no_grad():
        sky_c = renderer.sky_features(rd)
        no = fused.field_fused(renderer, vid, d2, rd, torch.from_numpy(g["cam_ori"]), sky_c,
                               torch.from_numpy(g["sky_avg"]).cuda().reshape(1, 64), ns, u=u.reshape(-1, ns + 1).cuda().contiguous())
    err = float(np.abs(no.view(1, hp, wp, 64).cpu().numpy() - ref.numpy()).max())
    det = float(np.abs(ref.numpy() - g["net_out"]).max())
    print(f"stochastic sampling, fused vs oracle: max abs err {err:.2e} (stochastic vs deterministic output differs by {det:.2e})")
    assert err < 1e-3 and det > 1e-3


def _frame(renderer, hw=(72, 104), pose_i=3):
    from scenedreamer_amd import camera
    poses = camera.eval_camera_poses(renderer.scene, maxstep=8)
    pose = poses[pose_i]
    vid, d2, rd, (H0, W0) = renderer.cast_rays(pose, hw)
    n = H0 * W0
    return pose, vid.view(n, renderer.M), d2.view(2, n, renderer.M), rd.view(n, 3), H0, W0


def test_sky_mean_finished_in_kernel(renderer):
    """The frame mean of the sky features is added up by sky_kernel's last workgroup (fixed order, double): equal to a
    float64 column mean of the sky_c it wrote, identical across launches (the arrival counter resets itself), and correct
    for ray counts that leave workgroups / waves partly or wholly idle."""
    from scenedreamer_amd import fused
    renderer.set_style_code(golden("field_a.npz")["z"])
    _, _, _, rd, _, _ = _frame(renderer)
    for n in (rd.shape[0], 33, 128 * 256 + 5):
        r = rd[:n] if n <= rd.shape[0] else torch.cat([rd] * (n // rd.shape[0] + 1))[:n]
        sky_c, avg = fused.sky_fused(renderer, r)
        ref = (sky_c.sum(dim=0, dtype=torch.float64) / n).to(torch.float32)
        assert (avg.reshape(-1) - ref).abs().max().item() < 1e-6
        sky_c2, avg2 = fused.sky_fused(renderer, r)
        assert torch.equal(avg, avg2) and torch.equal(sky_c, sky_c2)


def test_ray_window_equals_sliced_copies(renderer, monkeypatch):
    """encode / mlp reading the frame-wide ray arrays through a window (cropped apron, chunk offsets) produce the bits
    the same kernels produce on strided-slice COPIES of those rays; the dynamic group schedule (ticket counter)
    produces the bits of the static one.  (Row-major ray order: the copies have no window to take the blocked order from;
    test_blocked_ray_order_changes_no_bit covers that.)"""
    from scenedreamer_amd import fused
    monkeypatch.setenv("SDN_RAY_BLOCKS", "0")
    renderer.set_style_code(golden("field_a.npz")["z"])
    pose, vid, d2, rd, H0, W0 = _frame(renderer)
    ns, o, M = 12, 11, renderer.M
    sky_c, sky_avg = fused.sky_fused(renderer, rd)
    win = fused.Window.crop(H0, W0, o)
    # the copies the host side used to make
    cv = vid.view(H0, W0, M)[o:H0 - o, o:W0 - o].reshape(-1, M).contiguous()
    cd = d2.view(2, H0, W0, M)[:, o:H0 - o, o:W0 - o].reshape(2, -1, M).contiguous()
    cr = rd.view(H0, W0, 3)[o:H0 - o, o:W0 - o].reshape(-1, 3).contiguous()
    cs = sky_c.view(H0, W0, 64)[o:H0 - o, o:W0 - o].reshape(-1, 64).contiguous()
    assert win.n_rays == cv.shape[0]
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    b_copy = {k: v.clone() for k, v in fused.encode(renderer, cv, cd, cr, ori, ns).items()}
    b_win = fused.encode(renderer, vid, d2, rd, ori, ns, window=win)
    hit_rows = (cv[:, 0] != 0).view(-1, 1)
    for k in ("dist", "label", "rayflag"):
        assert torch.equal(b_copy[k], b_win[k]), k
    # features are only defined for tiles (8 rays) with a hit: compare those
    tiles = torch.nn.functional.pad(hit_rows.view(-1), (0, (-hit_rows.numel()) % 8)).view(-1, 8).any(dim=1)
    fa, fb = b_copy["feat"].view(tiles.numel(), -1)[tiles], b_win["feat"].view(tiles.numel(), -1)[tiles]
    assert torch.equal(fa, fb)
    st = renderer._fused_style or fused.prepare_style(renderer)
    outs = {}
    for tag, (buf, sc, w, dyn) in {"copy": (b_win, cs, None, False), "window": (b_win, sky_c, win, False),
                                   "dynamic": (b_win, sky_c, win, True), "dynamic2": (b_win, sky_c, win, True)}.items():
        no = torch.full((win.n_rays, 64), float("nan"), device="cuda")
        fused._launch_mlp(renderer, buf, st, sc, sky_avg.reshape(-1), no, win.n_rays, ns, window=w, dynamic=dyn)
        outs[tag] = no
    assert torch.equal(outs["copy"], outs["window"]) and torch.equal(outs["copy"], outs["dynamic"])
    assert torch.equal(outs["copy"], outs["dynamic2"]) and int(st["ticket"].abs().sum()) == 0     # the counter resets itself
    assert torch.isfinite(outs["copy"]).all()
    # chunked evaluation (ray0 offsets into the window) == one launch
    whole = fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win)
    assert torch.equal(whole, outs["copy"])
    old = fused.FEATURE_BUFFER_BYTES
    try:
        fused.FEATURE_BUFFER_BYTES = fused._per_ray_feat_bytes(ns) * 32 * 37     # 37 groups per chunk
        chunked = fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win)
    finally:
        fused.FEATURE_BUFFER_BYTES = old
    assert torch.equal(chunked, whole)


def test_blocked_ray_order_changes_no_bit(renderer, monkeypatch):
    """A whole-window launch of 8k columns x 4m rows takes its rays in 8 x 4 pixel blocks (RayWindow::pix, include/sdnative.h
    `window_host[5]`), so that the 32 rays of a group are neighbours in both directions.  Rays are independent: with every sample
    evaluated (term_eps = 0) net_out is the same bits as in row-major order -- one-kernel and two-kernel field, deterministic and
    stochastic sampling (u stays indexed by the pixel), the per-sample outputs -- and with early termination the two forms of
    the field still agree with each other bit for bit; a window that is not whole blocks stays row-major."""
    from scenedreamer_amd import fused
    renderer.set_style_code(golden("field_a.npz")["z"])
    pose, vid, d2, rd, H0, W0 = _frame(renderer)
    ns = 12
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    with torch.no_grad():
        sky_c, sky_avg = fused.sky_fused(renderer, rd)
        win = fused.Window.crop(H0, W0, 11)
        assert win.blocked(0, win.n_rays) and not fused.Window.crop(H0, W0, 10).blocked(0) and not win.blocked(32, win.n_rays - 32)
        torch.manual_seed(7)
        u = torch.rand(win.n_rays, ns + 1, device="cuda")
        try:
            outs = {}
            for blocks in ("0", "1"):
                monkeypatch.setenv("SDN_RAY_BLOCKS", blocks)
                for one in (True, False):
                    renderer.field_single_kernel = one
                    renderer.set_precision(term_eps=0.0)
                    a = fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win)
                    b = fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win, u=u)
                    renderer.set_precision(term_eps=0.05)
                    pa = torch.zeros((win.n_rays + 31) // 32, dtype=torch.uint8, device="cuda")
                    c = fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win, passes=pa)
                    outs[(blocks, one)] = (a, b, c, pa)
                renderer.set_precision(term_eps=0.0)
                aux = {"weights": None, "sigma": None, "nosky": None}
                fused.field_render(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win, aux=aux)
                outs[(blocks, "aux")] = aux
            for k in (0, 1):        # every sample evaluated: the same bits whatever the order and the form
                ref = outs[("0", True)][k]
                assert all(torch.equal(outs[(bl, one)][k], ref) for bl in ("0", "1") for one in (True, False)), k
            for bl in ("0", "1"):   # early termination: the two forms of the field agree under either order
                assert torch.equal(outs[(bl, True)][2], outs[(bl, False)][2]) and torch.equal(outs[(bl, True)][3], outs[(bl, False)][3])
            d = float((outs[("1", True)][2] - outs[("0", True)][2]).abs().max())
            assert d <= 2 * 0.05 + 1e-6                                                # different groups stop: inside the termination bound
            for k in ("weights", "sigma", "nosky"):
                assert torch.equal(outs[("0", "aux")][k], outs[("1", "aux")][k]), k
            assert not torch.equal(outs[("0", True)][0], outs[("0", True)][1])
        finally:
            renderer.field_single_kernel = None
            renderer.set_precision()


def test_ragged_blocked_ray_order_changes_no_bit(renderer, monkeypatch):
    """sdn_field_render on a whole window that is NOT whole 8 x 4 blocks (the frame itself: 102 x 134; a 82 x 114 crop; a 5 x 9 sliver):
    `window_host[5] == 2` walks the covering block grid, positions outside the window are no rays.  net_out, the per-sample outputs
    and nosky are the same bits as in row-major order (every sample evaluated); the per-group pass counters have one entry per
    block; a caller whose `passes` array is sized for row-major groups keeps the row-major order."""
    from scenedreamer_amd import fused
    renderer.set_style_code(golden("field_a.npz")["z"])
    pose, vid, d2, rd, H0, W0 = _frame(renderer)
    ns = 12
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    try:
        renderer.set_precision(term_eps=0.0)
        with torch.no_grad():
            sky_c, sky_avg = fused.sky_fused(renderer, rd)
            for win in (fused.Window.crop(H0, W0, 0), fused.Window.crop(H0, W0, 10), fused.Window(H0 * W0, W0, 40 * W0 + 60, 5, 9)):
                rows = win.n_rays // win.cols
                assert not win.blocked(0, win.n_rays) and win.blocked(0, win.n_rays, True) and (rows % 4 or win.cols % 8)
                monkeypatch.setenv("SDN_RAY_BLOCKS", "0")
                ref = fused.field_render(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win).clone()
                aux0 = {"weights": None, "sigma": None, "nosky": None, "colour": None}
                fused.field_render(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win, aux=aux0)
                monkeypatch.setenv("SDN_RAY_BLOCKS", "1")
                pa = torch.zeros(win.n_groups(True), dtype=torch.uint8, device="cuda")
                cp = torch.zeros_like(pa)
                got = fused.field_render(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win, passes=pa, colour_passes=cp)
                assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (rows, win.cols)
                aux1 = {"weights": None, "sigma": None, "nosky": None, "colour": None}
                fused.field_render(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win, aux=aux1)
                for k in aux0:
                    assert torch.equal(aux0[k], aux1[k]), k
                # the counters are per block: a block is visited iff one of its rays hits, and then goes through every pass
                hit = (vid.view(H0, W0, -1)[..., 0] != 0).reshape(-1)
                first, pitch = win.first, win.pitch
                yy, xx = torch.meshgrid(torch.arange(rows, device="cuda"), torch.arange(win.cols, device="cuda"), indexing="ij")
                hit_w = hit[(first + yy * pitch + xx).reshape(-1)]
                visited = win.groups(hit_w, ragged=True).any(dim=1)
                assert torch.equal(pa > 0, visited) and bool((pa[visited] == -(-ns // 4)).all()) and bool((cp <= pa).all())
                # a `passes` array sized for the row-major groups: the launch stays row-major (and is still the same bits)
                small = torch.zeros((win.n_rays + 31) // 32, dtype=torch.uint8, device="cuda")
                if small.numel() < win.n_groups(True):
                    again = fused.field_render(renderer, vid, d2, rd, ori, sky_c, sky_avg, ns, window=win, passes=small)
                    assert torch.equal(again, ref)
    finally:
        renderer.set_precision()


# ---------------------------------------------------------------------------------------------------- MX fp6 colour layers
def test_mx_fp6_hardware_facts():
    """tools/mx_probe (built by __graft_entry__.build): element order / scale semantics / rounding of the fp6 conversions and
    the operand layout + per-lane E8M0 scale bytes of v_mfma_scale_f32_32x32x64_f8f6f4, as layer8x and pack_mx_kernel assume."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mx_probe")
    if not os.path.exists(exe):
        pytest.skip("tools/mx_probe not built")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    lines = out.splitlines()
    nat = " ".join(str(i) for i in range(32)).replace(" 5 ", " 37 ")          # element 5 is negative in the probe
    inter = " ".join(str(v) for i in range(16) for v in (i, 16 + i)).replace(" 5 ", " 37 ")
    for sc in ("1", "2", "0.5", "3"):
        assert any(l.startswith(f"cvt pk32_fp6_f16   scale {sc} ") and l.rstrip().endswith(nat) for l in lines), sc
        assert any(l.startswith(f"cvt 2xpk16_fp6_f32 scale {sc} ") and l.rstrip().endswith(inter) for l in lines), sc
    assert sum("wrong codes pk32_fp6_f16 0, 2xpk16_fp6_f32 0" in l for l in lines) == 4
    assert sum(l.startswith("mfma_scale fp6 test") and l.rstrip().endswith("OK") for l in lines) == 6 and "MISMATCH" not in out
    # round to nearest even, saturating at 7.5 (element i of the awkward inputs sits at position 2 i)
    rnd = next(l for l in lines if l.startswith("rounding"))
    got = {int(t[1:t.index("]")]): float(t.split("=")[1]) for t in rnd.split() if t.startswith("[")}
    want = [7.5, 7.5, 7.5, 7.5, 7.5, 0.0, 0.125, 0.25, 0.25, 3.75, 4.0, 1.0, 1.125, -7.5, 0.0, 6.0]
    assert [got[2 * i] for i in range(16)] == want


def test_mx_weight_image_decodes_to_the_weights(renderer):
    """pack_mx_kernel: the colour layers' part of the packed weight stream (f16 Whi fragments of a k-step for four row
    blocks; fp6 fragments of Wlo and Whi with one E8M0 scale per row and 32-k block) decoded on the host equals the folded
    weights: Whi exactly, the fp6 parts to half an fp6 step of the block maximum."""
    from scenedreamer_amd import fused
    renderer.set_style_code(golden("field_a.npz")["z"])
    st = fused.prepare_style(renderer)
    torch.cuda.synchronize()
    raw = st["packed_mx"].cpu().numpy().view(np.uint32)
    L0, LH = 8 * 8 * 2 * 64, 16 * 8 * 2 * 64                     # fragments (16 bytes per lane) of fc_1 / of a hidden layer

    def dec6(c):
        c = np.asarray(c, np.int64)
        sg, e, m = (c >> 5) & 1, (c >> 3) & 3, c & 7
        v = np.where(e == 0, m * 0.125, (1 + m * 0.125) * np.exp2(e - 1.0))
        return np.where(sg == 1, -v, v)

    kmap = lambda s_, h, e: 32 * (s_ >> 1) + 16 * (s_ & 1) + (e & 3) + 8 * (e >> 2) + 4 * h
    worst = {0: 0.0, 1: 0.0}
    for layer, name in ((0, 5), (1, 6)):
        W = renderer.mod[name][0].cpu().numpy().astype(np.float32) * np.float32(0.4)
        Whi = W.astype(np.float16).astype(np.float32)
        Wlo = W - Whi
        base = (L0 + (3 + layer) * LH) * 4
        for u in range(0, 64, 3):
            half, kb, sub, ib0 = u // 32, (u % 32) // 8, u % 8, 4 * (u // 32)
            ub = base + u * 4 * 64 * 4
            for lane in (0, 5, 31, 32, 63):
                h = lane >> 5
                frag = lambda f: raw[ub + (f * 64 + lane) * 4: ub + (f * 64 + lane) * 4 + 4]
                if sub < 4:
                    for f in range(4):
                        row = 32 * (ib0 + f) + (lane & 31)
                        ref = np.array([Whi[row, kmap(4 * kb + sub, h, e)] for e in range(8)])
                        np.testing.assert_array_equal(frag(f).view(np.float16).astype(np.float32), ref)
                    continue
                term, iba = (sub - 4) // 2, ib0 + 2 * ((sub - 4) % 2)
                for rb in range(2):
                    f0, f1 = frag(2 * rb), frag(2 * rb + 1)
                    big = sum(int(v) << (32 * i) for i, v in enumerate(list(f0) + list(f1[:2])))
                    val = dec6([(big >> (6 * i)) & 63 for i in range(32)]) * np.exp2(float(int(f1[2]) - 127))
                    row = 32 * (iba + rb) + (lane & 31)
                    ref = np.array([(Wlo if term == 0 else Whi)[row, kmap(4 * kb + i // 8, h, i % 8)] for i in range(32)])
                    # (the sign of an exact f16 rounding tie of Wlo is the device's; magnitudes are compared)
                    err = np.abs(np.abs(val) - np.abs(ref)).max() / max(np.abs(ref).max(), 1e-30)
                    worst[term] = max(worst[term], float(err))
    assert worst[0] < 0.07 and worst[1] < 0.07, worst            # fp6 step in the top binade: 0.5 / 7.5 of the block maximum


def test_single_kernel_field_equals_the_two_kernel_field(renderer):
    """sdn_field_render (field_kernel: every pass places its samples and gathers its features itself, then runs the MLP on
    them) against sdn_field_encode + sdn_field_mlp: the same device functions produce the features either way, so net_out,
    the per-group pass counts and the stochastic-sampling variant must agree bit for bit -- whole ray set, cropped window,
    both precision profiles."""
    from scenedreamer_amd import fused
    pose, vid, d2, rd, H0, W0 = _frame(renderer)
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    with torch.no_grad():
        sky_c, sky_avg = fused.sky_fused(renderer, rd)
        n = vid.shape[0]
        torch.manual_seed(3)
        u = torch.rand(n, 13, device="cuda")
        win = fused.Window.crop(H0, W0, 11)
        uw = u[:win.n_rays].contiguous()
        try:
            for ct in (6, 3):
                renderer.set_precision(colour_terms=ct)
                outs = {}
                for one in (False, True):
                    renderer.field_single_kernel = one
                    pa = torch.zeros((n + 31) // 32, dtype=torch.uint8, device="cuda")
                    a = fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, 12, passes=pa)
                    b = fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, 12, window=win)
                    c = fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, 12, u=uw, window=win)
                    outs[one] = (a, b, c, pa)
                for x, y in zip(outs[False], outs[True]):
                    assert x.shape == y.shape and torch.equal(x, y)
                assert float(outs[True][0].std()) > 1e-2 and int(outs[True][3].sum()) > 0
                assert not torch.equal(outs[True][1], outs[True][2])          # (the stochastic draw does change the result)
        finally:
            renderer.field_single_kernel = None
            renderer.set_precision()


def test_colour_branch_skipping_is_bit_exact(renderer):
    """field_kernel leaves out fc_5 / fc_6 / fc_out_c in passes whose 128 samples all have relu(sigma) * dist == 0 (weights
    exactly zero, mc_utils.py:154-161): net_out must not change by one bit against the launch that evaluates every pass --
    whole ray set, cropped window, both precision profiles, with and without early termination -- and the skipped passes are
    counted (colour_passes <= passes, equal where nothing was skipped)."""
    from scenedreamer_amd import fused
    pose, vid, d2, rd, H0, W0 = _frame(renderer)
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    n = vid.shape[0]
    ng = (n + 31) // 32
    with torch.no_grad():
        sky_c, sky_avg = fused.sky_fused(renderer, rd)
        win = fused.Window.crop(H0, W0, 7)
        try:
            renderer.field_single_kernel = True
            for ct, eps in ((6, 0.0), (3, 0.0), (6, 5e-5), (6, 0.2)):
                renderer.set_precision(colour_terms=ct, term_eps=eps)
                outs = {}
                for skip in (False, True):
                    renderer.colour_skip = skip
                    pa, cp = torch.zeros(ng, dtype=torch.uint8, device="cuda"), torch.full((ng,), 255, dtype=torch.uint8, device="cuda")
                    a = fused.field_render(renderer, vid, d2, rd, ori, sky_c, sky_avg, 24, passes=pa, colour_passes=cp)
                    b = fused.field_render(renderer, vid, d2, rd, ori, sky_c, sky_avg, 24, window=win)
                    outs[skip] = (a, b, pa, cp)
                assert torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1]), (ct, eps)
                assert torch.equal(outs[False][2], outs[True][2])                       # the same passes were gone through
                assert torch.equal(outs[False][3], outs[False][2])                      # no skipping: every pass ran the colour branch
                assert bool((outs[True][3] <= outs[True][2]).all())
                ran, went = int(outs[True][3].sum(dtype=torch.int64)), int(outs[True][2].sum(dtype=torch.int64))
                print(f"colour_terms {ct}, term_eps {eps}: colour branch evaluated in {ran} of {went} passes ({100 * (1 - ran / max(went, 1)):.1f} % skipped)")
                assert 0 < ran < went                                                     # the synthetic field does have empty space
        finally:
            renderer.field_single_kernel = None
            renderer.colour_skip = None
            renderer.set_precision()


def test_early_termination_single_kernel_equals_two_kernel(renderer):
    """term_eps > 0 (wavefront-ballot early ray termination): the single-kernel field finishes `is_gnd` over the passes it
    skips (placement only), so it still chooses the same sky term as encode_kernel -> mlp_kernel, which knows all samples up
    front -- same bits, same pass counts; and net_out moves by at most 2 * term_eps against the untruncated evaluation."""
    from scenedreamer_amd import fused
    pose, vid, d2, rd, H0, W0 = _frame(renderer)
    ori = torch.as_tensor(pose[0], dtype=torch.float32)
    n = vid.shape[0]
    with torch.no_grad():
        sky_c, sky_avg = fused.sky_fused(renderer, rd)
        try:
            renderer.set_precision()
            full = fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, 24)
            # an opaque variant of the density head, so that rays do terminate on the synthetic weights
            for eps in (1e-3, 0.3):
                outs = {}
                for one in (False, True):
                    renderer.set_precision(term_eps=eps)
                    renderer.field_single_kernel = one
                    pa = torch.zeros((n + 31) // 32, dtype=torch.uint8, device="cuda")
                    outs[one] = (fused.field_fused(renderer, vid, d2, rd, ori, sky_c, sky_avg, 24, passes=pa), pa)
                assert torch.equal(outs[False][1], outs[True][1]) and torch.equal(outs[False][0], outs[True][0]), eps
                skipped = int((outs[True][1] < 6).sum()) - int((outs[True][1] == 0).sum())
                d = float((outs[True][0] - full).abs().max())
                print(f"term_eps {eps}: {skipped} of {int((outs[True][1] > 0).sum())} visited groups stopped early; max |net_out - untruncated| {d:.2e}")
                assert d <= 2 * eps + 1e-6
        finally:
            renderer.field_single_kernel = None
            renderer.set_precision()
