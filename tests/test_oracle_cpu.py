"""CPU-only checks that pin the ORACLE itself: golden vectors recorded from the unmodified reference Python,
analytic known-answer tests, and an independent numpy formulation of the hash-grid encoder."""
import numpy as np
import pytest
import torch

from conftest import bits, golden


# ------------------------------------------------------------------ golden vectors from the reference Python
def test_field_ref_reproduces_reference_golden(oracle, weights_full, scene256, lut):
    """oracle/field_ref.py == imaginaire.generators.scenedreamer.Generator._forward_perpix/_forward_global
    (recorded by oracle/make_golden.py) bit for bit on CPU."""
    from oracle import field_ref as FR
    for tag in "abc":
        g = golden(f"field_{tag}.npz")
        no, aux = FR.forward_perpix(weights_full, lut, scene256.voxel_t.shape, g["voxel_id"], g["depth2"], g["raydirs"],
                                    g["cam_ori"][None], g["z"], g["global_enc"], int(g["num_samples"]),
                                    sky_avg=g["sky_avg"], return_aux=True)
        img = FR.render_cnn(weights_full, no, g["z"])
        np.testing.assert_array_equal(aux["new_idx"].numpy().astype(np.int8), g["new_idx"])
        np.testing.assert_array_equal(bits(aux["rand_depth"].numpy()), bits(g["rand_depth"]))
        np.testing.assert_allclose(no.numpy(), g["net_out"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(img.numpy(), g["image"], rtol=0, atol=1e-6)


def test_zero_weight_samples_do_not_reach_net_out(oracle, weights_full, scene256, lut):
    """The property field_kernel's colour-branch skipping rests on, stated on the reference's own arithmetic (the oracle's
    restatement of mc_utils.py:154-161 and scenedreamer.py:373-413): a sample with relu(sigma) * dist == 0 gets volume-rendering
    weight EXACTLY 0.0, so net_out is the same bits whatever its colour is -- the kernel may leave fc_5 / fc_6 / fc_out_c out for
    a pass made of such samples only.  Checked on the goldens' inputs: the weights are exact zeros there, and recompositing with
    those samples' colours replaced (by huge values, by zero) reproduces net_out bit for bit."""
    from oracle import field_ref as FR
    seen = 0
    for tag in "abc":
        g = golden(f"field_{tag}.npz")
        no, aux = FR.forward_perpix(weights_full, lut, scene256.voxel_t.shape, g["voxel_id"], g["depth2"], g["raydirs"],
                                    g["cam_ori"][None], g["z"], g["global_enc"], int(g["num_samples"]),
                                    sky_avg=g["sky_avg"], return_aux=True)
        sigma, dists, wts, col, sky = aux["sigma"], aux["new_dists"] * 0.25, aux["weights"], aux["color"], aux["sky"]
        empty = (torch.relu(sigma) * dists) == 0
        assert bool((wts[empty] == 0).all()) and bool(empty.any()) and not bool(empty.all())
        seen += int(empty.sum())

        def composite(colour):        # scenedreamer.py:407-413 on the oracle's per-sample outputs
            rgbs = torch.clamp(colour, -1, 1) + 1
            rgbs_sky = torch.clamp(sky, -1, 1) + 1
            return (torch.sum(wts * rgbs, dim=-2, keepdim=True) + (1.0 - torch.sum(wts, dim=-2, keepdim=True)) * rgbs_sky).squeeze(-2) - 1
        base = composite(col)
        np.testing.assert_array_equal(bits(base.numpy()), bits(no.numpy()))
        for junk in (1e30, 0.0, -7.0):
            other = torch.where(empty.expand_as(col), torch.full_like(col, junk), col)
            np.testing.assert_array_equal(bits(composite(other).numpy()), bits(no.numpy()))
    assert seen > 1000


def test_sample_depth_oracle_reproduces_reference_golden():
    """oracle/field_ref.sample_depth_batched (deterministic and the training-time stochastic branch, fed with the stored
    torch.rand draw) == the unmodified mc_utils.sample_depth_batched (oracle/make_golden_sampling.py), bit for bit."""
    from oracle import field_ref as FR
    g = golden("stochastic_sampling.npz")
    d2 = torch.from_numpy(g["depth2"])
    for ns in (13, 25):
        rd, nd, idx = FR.sample_depth_batched(d2.clone(), ns, 3.0, rand=g[f"u{ns}"])
        np.testing.assert_array_equal(idx.numpy().astype(np.int8), g[f"idx{ns}"])
        np.testing.assert_array_equal(bits(rd.numpy()), bits(g[f"depth{ns}"]))
        np.testing.assert_array_equal(bits(nd.numpy()), bits(g[f"dists{ns}"]))
        rd, nd, idx = FR.sample_depth_batched(d2.clone(), ns, 3.0)
        np.testing.assert_array_equal(idx.numpy().astype(np.int8), g[f"det_idx{ns}"])
        np.testing.assert_array_equal(bits(rd.numpy()), bits(g[f"det_depth{ns}"]))


def test_style_and_scene_codes_reproduce_reference_golden(weights_full, scene256):
    from oracle import field_ref as FR
    from scenedreamer_amd import synth
    g = golden("style_globalenc.npz")
    z = FR.style_mlp(weights_full, synth.make_style(int(g["z_seed"])))
    ge = FR.world_encoder(weights_full, scene256.current_height_map, scene256.current_semantic_map)
    np.testing.assert_allclose(z.numpy(), g["z"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ge.numpy(), g["global_enc"], rtol=0, atol=1e-6)


def test_rvip_oracle_reproduces_golden_rays(oracle, scene256):
    """The C ray marcher regenerates the intersections stored with the golden field vectors."""
    for tag in "abc":
        g = golden(f"field_{tag}.npz")
        hw = [int(v) + 30 for v in g["resolution_hw"]]
        vid, d2, rd = oracle.rvip(scene256.voxel_t.numpy(), g["cam_ori"], g["cam_dir"], g["cam_up"], float(g["cam_f"]),
                                  g["cam_c"], hw, 6)
        np.testing.assert_array_equal(vid[None], g["voxel_id"])
        np.testing.assert_array_equal(bits(d2[None]), bits(g["depth2"]))
        np.testing.assert_array_equal(bits(rd[None]), bits(g["raydirs"]))


# ------------------------------------------------------------------ analytic known answers
def test_rvip_single_voxel_world(oracle):
    vox = np.zeros((8, 8, 8), np.int32)
    vox[4, 4, 4] = 7
    ids, d2, dirs, steps = oracle.rvip(vox, [4.5, 4.5, 0.5], [0, 0, 1], [1, 0, 0], 1.0, [0.0, 0.0], [1, 1], 3, True)
    assert ids.ravel().tolist() == [7, 0, 0]
    assert d2.ravel()[0] == 3.5 and d2.ravel()[3] == 4.5            # entry at z=4, exit at z=5, origin z=0.5
    assert np.isnan(d2.ravel()[[1, 2, 4, 5]]).all()
    assert dirs.ravel().tolist() == [0.0, 0.0, 1.0]
    assert steps[0, 0] == 8                                          # z cells 1..7 then the step that leaves


def test_rvip_origin_cell_is_never_tested(oracle):
    vox = np.full((4, 4, 4), 5, np.int32)
    ids, d2, _ = oracle.rvip(vox, [1.5, 1.5, 1.5], [0, 0, 1], [1, 0, 0], 1.0, [0.0, 0.0], [1, 1], 4)
    assert ids.ravel().tolist() == [5, 5, 0, 0]                      # cells z=2,3; the origin cell z=1 is skipped
    np.testing.assert_array_equal(d2.ravel()[:2], [0.5, 1.5])


def test_rvip_pure_python_reference(oracle):
    """Independent per-ray restatement in float32 numpy scalars (tiny image)."""
    rng = np.random.default_rng(0)
    vox = (rng.random((6, 9, 9)) < 0.2).astype(np.int32) * rng.integers(1, 50, (6, 9, 9)).astype(np.int32)
    ori, cdir, up = np.float32([2.3, -3.0, 4.1]), np.float32([0.1, 1.0, 0.05]), np.float32([1, 0, 0])
    f, c, dims, M = np.float32(9.0), np.float32([3.5, 4.5]), [8, 10], 3
    ids, d2, dirs = oracle.rvip(vox, ori, cdir, up, f, c, dims, M)
    fwd, side, upv = oracle.camera_frame(cdir, up)
    F = np.float32
    for r in range(dims[0]):
        for cc in range(dims[1]):
            n0, n1 = F(c[0] - F(r)), F(F(cc) - c[1])
            d = np.array([F(F(F(upv[i] * n0) + F(side[i] * n1)) + F(fwd[i] * f)) for i in range(3)], np.float32)
            ln = np.sqrt(F(F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2])), dtype=np.float32)
            d = (d / ln).astype(np.float32)
            np.testing.assert_array_equal(dirs[r, cc, 0], d)
            cell = np.floor(ori).astype(np.int64)
            with np.errstate(divide="ignore", invalid="ignore"):
                t = np.array([F(F(F(cell[i] + (1 if d[i] > 0 else 0)) - ori[i]) / d[i]) if d[i] != 0 else np.inf
                              for i in range(3)], np.float32)
            out, quit_ = [], False
            for _ in range(M):
                hit = (np.nan, np.nan, 0)
                while not quit_:
                    a = 0 if (t[0] <= t[1] and t[0] <= t[2]) else (1 if t[1] <= t[2] else 2)
                    tnow = t[a]
                    cell[a] += 1 if d[a] > 0 else -1
                    quit_ = cell[a] >= vox.shape[a] if d[a] > 0 else cell[a] < 0
                    t[a] = F(F(F(cell[a] + (1 if d[a] > 0 else 0)) - ori[a]) / d[a])
                    if quit_:
                        break
                    if (cell < 0).any() or (cell >= np.array(vox.shape)).any():
                        continue
                    b = vox[tuple(cell)]
                    if b == 0:
                        continue
                    hit = (tnow, t.min(), int(b))
                    break
                out.append(hit)
            assert [h[2] for h in out] == ids[r, cc, :, 0].tolist()
            np.testing.assert_array_equal(bits(np.float32([h[0] for h in out])), bits(d2[0, r, cc, :, 0]))
            np.testing.assert_array_equal(bits(np.float32([h[1] for h in out])), bits(d2[1, r, cc, :, 0]))


def test_hash_and_level_constants(oracle):
    P = [1, 2654435761, 805459861, 3674653429, 2097192037]
    pg = [3, 77, 1200, 5, 9]
    h = 0
    for a, b in zip(pg, P):
        h ^= (a * b) & 0xFFFFFFFF
    assert oracle.fast_hash(pg) == h
    S = np.float32(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    # per-level scale as the kernel computes it in fp32
    expect = {0: (15.0, 16), 1: (21.1106, 23), 4: (57.3502, 59), 8: (211.797, 213), 12: (775.047, 777), 15: (2047.0, 2048)}
    for l, (sc, res) in expect.items():
        s, r = oracle.level_params(l, S, 16)
        assert abs(s - sc) < 2e-3 and r == res
    # every level of the SceneDreamer grid takes the hash branch (stride (res+1)^5 > 2^19)
    for l in range(16):
        _, r = oracle.level_params(l, S, 16)
        assert (r + 1) ** 5 > 2 ** 19
        assert oracle.grid_index(5, 8, 0, False, 2 ** 19, r, pg) == (h % 2 ** 19) * 8
    # dense (tiled) indexing when the grid fits
    assert oracle.grid_index(2, 2, 0, False, 4096, 15, [3, 5]) == (3 + 5 * 16) * 2


def test_grid_encode_constant_table_and_oob(oracle):
    from scenedreamer_amd.gridencoder import level_offsets
    offs = level_offsets(3, 5, 1.5, 4, 10, False)
    emb = np.full((int(offs[-1]), 4), 0.25, np.float32)
    x = np.random.default_rng(0).random((200, 3), dtype=np.float32)
    x[0] = [0, 0, 0]
    x[1] = [1, 1, 1]
    x[2] = [0.5, -1e-6, 0.5]
    x[3] = [0.5, 0.5, 1.0001]
    out = oracle.grid_encode_fwd(x, emb, offs, np.float32(np.log2(1.5)), 4)
    assert (out[:, 2:4] == 0).all()                                   # out of [0,1] -> zeros (gridencoder.cu:99-123)
    np.testing.assert_allclose(out[:, [0, 1] + list(range(4, 200))], 0.25, atol=1e-6)   # weights sum to 1


def test_grid_encode_independent_numpy_formulation(oracle):
    """Vectorised numpy restatement (all samples at once, corner loop outermost) vs the scalar C oracle."""
    from scenedreamer_amd.gridencoder import level_offsets
    rng = np.random.default_rng(7)
    P = np.array([1, 2654435761, 805459861, 3674653429, 2097192037], np.uint64)
    for D, C, T in ((5, 8, 12), (3, 2, 9), (2, 4, 14)):
        L, H, pls = 6, 4, 1.7
        offs = level_offsets(D, L, pls, H, T, False)
        emb = rng.random((int(offs[-1]), C), dtype=np.float32) - 0.5
        x = rng.random((500, D), dtype=np.float32)
        S = np.float32(np.log2(pls))
        ref = oracle.grid_encode_fwd(x, emb, offs, S, H)
        for l in range(L):
            scale, res = oracle.level_params(l, S, H)
            size = int(offs[l + 1] - offs[l])
            pos = (x * np.float32(scale)).astype(np.float32) + np.float32(0.5)
            pg = np.floor(pos).astype(np.uint64)
            fr = (pos - pg.astype(np.float32)).astype(np.float32)
            acc = np.zeros((x.shape[0], C), np.float32)
            for corner in range(1 << D):
                w = np.ones(x.shape[0], np.float32)
                idx_dense = np.zeros(x.shape[0], np.uint64)
                hsh = np.zeros(x.shape[0], np.uint64)
                stride = 1
                for d in range(D):
                    bit = (corner >> d) & 1
                    w = w * (fr[:, d] if bit else (np.float32(1) - fr[:, d]))
                    pd = pg[:, d] + np.uint64(bit)
                    hsh ^= (pd * P[d]) & np.uint64(0xFFFFFFFF)
                    if stride <= size:
                        idx_dense = (idx_dense + pd * np.uint64(stride)) & np.uint64(0xFFFFFFFF)
                        stride *= res + 1
                index = hsh if stride > size else idx_dense
                acc += w[:, None] * emb[int(offs[l]) + (index % np.uint64(size)).astype(np.int64)]
            np.testing.assert_allclose(acc, ref[l], rtol=0, atol=2e-6)


def test_posenc_matches_reference_torch_twin(oracle):
    """positional_encoding_pt (positional_encoding.py:45-54) is the reference's own CPU twin of the kernel."""
    x = torch.rand(17, 5, 3) * 2 - 1
    for ndeg, incl in ((5, True), (4, False)):
        twin = torch.cat([fn(x * np.pi * 2 ** i) for i in range(ndeg) for fn in (torch.sin, torch.cos)] + ([x] if incl else []), dim=-1)
        np.testing.assert_allclose(oracle.posenc_fwd(x.numpy(), ndeg, -1, incl), twin.numpy(), rtol=1e-5, atol=1e-5)
    g = np.random.default_rng(1).standard_normal((17, 5, 33)).astype(np.float32)
    xr = x.clone().requires_grad_(True)
    y = torch.cat([fn(xr * np.pi * 2 ** i) for i in range(5) for fn in (torch.sin, torch.cos)] + [xr], dim=-1)
    y.backward(torch.from_numpy(g))
    np.testing.assert_allclose(oracle.posenc_bwd(g, oracle.posenc_fwd(x.numpy(), 5, -1, True), 5, -1, True), xr.grad.numpy(),
                               rtol=1e-4, atol=1e-4)


def test_grid_backward_oracle_against_autograd(oracle):
    """grid_encode_bwd oracle vs torch autograd through a differentiable numpy/torch re-expression (3-D, dense level)."""
    from scenedreamer_amd.gridencoder import level_offsets
    offs = level_offsets(2, 1, 2.0, 4, 12, False)          # one dense 5x5 level
    rng = np.random.default_rng(3)
    emb = torch.tensor(rng.random((int(offs[-1]), 2), dtype=np.float32) - 0.5, requires_grad=True)
    x = torch.tensor(rng.random((50, 2), dtype=np.float32), requires_grad=True)
    scale = 3.0
    pos = x * scale + 0.5
    pg = torch.floor(pos).detach()
    fr = pos - pg
    out = 0
    for cx in (0, 1):
        for cy in (0, 1):
            w = (fr[:, 0] if cx else 1 - fr[:, 0]) * (fr[:, 1] if cy else 1 - fr[:, 1])
            idx = ((pg[:, 0] + cx) + (pg[:, 1] + cy) * 5).long()
            out = out + w[:, None] * emb[idx]
    g = torch.tensor(rng.standard_normal((50, 2)).astype(np.float32))
    out.backward(g)
    S = np.float32(0.0)
    fwd, dy_dx = oracle.grid_encode_fwd(x.detach().numpy(), emb.detach().numpy(), offs, S, 4, True)
    np.testing.assert_allclose(fwd[0], out.detach().numpy(), atol=1e-6)
    gg, gi = oracle.grid_encode_bwd(g.numpy()[None], x.detach().numpy(), tuple(emb.shape), offs, S, 4, dy_dx)
    np.testing.assert_allclose(gg, emb.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(gi, x.grad.numpy(), atol=1e-4)
