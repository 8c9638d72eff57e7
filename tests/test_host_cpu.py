"""Host-side logic that needs no GPU: camera trajectory, tile arithmetic, synthetic data determinism,
the C ABI's export table and argument validation."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden


def test_camera_poses_match_reference_controller(scene256):
    """camera.eval_camera_poses == EvalCameraController(pattern=0) recorded from the reference (camctl.py:20-50)."""
    from scenedreamer_amd import camera
    g = golden("camera_pattern0.npz")
    poses = camera.eval_camera_poses(scene256, maxstep=int(g["maxstep"]))
    assert len(poses) == 8
    for i, (o, d, u, f) in enumerate(poses):
        np.testing.assert_array_equal(o.numpy(), g["ori"][i])
        np.testing.assert_array_equal(d.numpy(), g["dir"][i])
        np.testing.assert_array_equal(u.numpy(), g["up"][i])
        assert f == g["f"][i]


def test_all_ten_camera_patterns_match_reference(scene256):
    from scenedreamer_amd import camera
    g = golden("camera_patterns.npz")
    for pat in range(10):
        poses = camera.eval_camera_poses(scene256, maxstep=int(g["maxstep"]), pattern=pat)
        for i, (o, d, u, f) in enumerate(poses):
            np.testing.assert_array_equal(o.numpy(), g[f"ori{pat}"][i])
            np.testing.assert_array_equal(d.numpy(), g[f"dir{pat}"][i])
            np.testing.assert_array_equal(u.numpy(), g[f"up{pat}"][i])
            assert float(f) == g[f"f{pat}"][i]
    with pytest.raises(ValueError):
        camera.eval_camera_poses(scene256, maxstep=4, pattern=10)


@pytest.mark.needs_reference
def test_camera_poses_match_live_reference(scene256):
    from oracle import ref_harness as RH
    RH.install("oracle")
    import imaginaire.model_utils.gancraft.camctl as camctl
    from scenedreamer_amd import camera
    for pat in range(10):
        ctl = camctl.EvalCameraController(scene256, maxstep=12, pattern=pat, cam_ang=72, smooth_decay_multiplier=150 / 12)
        mine = camera.eval_camera_poses(scene256, maxstep=12, pattern=pat)
        for a, b in zip(ctl, mine):
            for x, y in zip(a[:3], b[:3]):
                np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
            assert float(a[3]) == float(b[3])


def test_tile_grid_and_intrinsics():
    from scenedreamer_amd import camera
    f, c, cam_res = camera.frame_intrinsics(0.5 / np.tan(np.deg2rad(36)), (540, 960), 30)
    assert cam_res == [570, 990] and c == [284.5, 494.5]
    assert abs(f - 0.5 / np.tan(np.deg2rad(36)) * 959) < 1e-9
    tiles, nh, nw = camera.tile_grid(cam_res, 30)
    assert (nh, nw, len(tiles)) == (5, 8, 40)
    assert sum((a[1] - a[0]) * (a[3] - a[2]) for a in tiles) == 828000
    tiles3, nh3, nw3 = camera.tile_grid([1110, 1950], 30)
    assert (nh3, nw3) == (9, 15) and sum((a[1] - a[0]) * (a[3] - a[2]) for a in tiles3) == 3199500


def test_level_offsets_of_the_scenedreamer_grid():
    from scenedreamer_amd.gridencoder import level_offsets
    pls = np.exp2(np.log2(2048 / 16) / 15)
    offs = level_offsets(5, 16, pls, 16, 19, False)
    assert offs.tolist() == [524288 * i for i in range(17)]
    small = level_offsets(2, 4, 2.0, 4, 19, False)
    assert small.tolist() == [0, 32, 32 + 88, 32 + 88 + 296, 32 + 88 + 296 + 1096]   # (res+1)^2 rounded up to 8


def test_synthetic_data_is_deterministic_and_well_formed(scene256, lut):
    from scenedreamer_amd import synth
    u = synth.hash_u01(5, "x", 4)
    np.testing.assert_array_equal(u, synth.hash_u01(5, "x", 6)[:4])
    assert not np.array_equal(u, synth.hash_u01(6, "x", 4))
    np.testing.assert_array_equal(synth.uniform(1, "a", (7,), -1, 1), synth.uniform(1, "a", (7,), -1, 1))
    s2 = synth.make_scene(256, 3407)
    assert torch.equal(s2.voxel_t, scene256.voxel_t)
    v = scene256.voxel_t
    assert v.dtype == torch.int32 and v.dim() == 3 and v.shape[1:] == (256, 256)
    ids = set(np.unique(v.numpy()).tolist())
    assert ids <= {0, 1, 8, 9, 26, 28, 30, 34, 58}
    assert all(lut[i] != 0 or i == 0 for i in ids)                  # every id has a reduced label
    assert lut[34] == 2 and lut[58] == 2                             # tree blocks
    assert scene256.current_semantic_map.shape == (1, 11, 256, 256)
    assert scene256.current_semantic_map[0, 10].sum() > 0            # >= 1 tree pixel (layers.py:28 needs 11 channels)
    assert float(scene256.trans_mat[0, 3]) == float(scene256.heightmap.min())
    w = synth.make_weights(0, with_embeddings=False)
    assert w["render_net.fc_1.weight"].shape == (256, 128) and w["render_net.fc_m_a.weight"].shape == (256, 12)
    assert w["denoiser.conv2a.weight"].shape == (256, 256, 3, 3) and "denoiser.conv2b.bias" not in w


def test_label_lut():
    from scenedreamer_amd.renderer import load_label_lut
    d = load_label_lut()
    assert len(d["lut"]) == 680 and d["num_reduced"] == 12 and d["ignore_id"] == 0 and d["dirt_id"] == 3
    assert d["lut"][0] == 1                                           # air -> sky
    assert [d["lut"][i] for i in (28, 9, 8, 1, 30, 26)] == [10, 3, 5, 9, 6, 7]


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "sdnative.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from scenedreamer_amd import capi
    lib = capi.lib()
    declared = _header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sdnative.h but not exported"
    assert set(capi.declared_symbols()) <= set(declared)
    assert lib.sdn_abi_version() == capi.ABI_VERSION == 5


def _header_prototypes():
    """name -> list of C parameter declarations, parsed from include/sdnative.h."""
    txt = open(os.path.join(ROOT, "include", "sdnative.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    protos = {}
    for m in re.finditer(r"\b(sdn_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        args = " ".join(m.group(2).split())
        protos[m.group(1)] = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
    return protos


def test_ctypes_signatures_match_the_header():
    """The ctypes binding (capi.py) and include/sdnative.h must agree on every entry point: parameter count, and
    pointer / integer / float class of each parameter (an ABI drift would otherwise only show up as garbage on the GPU)."""
    from scenedreamer_amd import capi
    protos = _header_prototypes()
    sigs = dict(capi._SIGNATURES)
    sigs.update(capi.EXTRA_SIGNATURES)
    assert set(sigs) == set(protos), sorted(set(sigs) ^ set(protos))
    for name, (_, argtypes) in sigs.items():
        decl = protos[name]
        assert len(decl) == len(argtypes), f"{name}: header has {len(decl)} parameters, ctypes {len(argtypes)}"
        for d, t in zip(decl, argtypes):
            is_ptr_c = "*" in d or "sdn_stream_t" in d
            is_float_c = (not is_ptr_c) and re.search(r"\bfloat\b|\bdouble\b", d) is not None
            is_ptr_py = t is ctypes.c_void_p
            is_float_py = t in (ctypes.c_float, ctypes.c_double)
            assert is_ptr_c == is_ptr_py and is_float_c == is_float_py, f"{name}: `{d}` bound as {t.__name__}"


def test_hand_counted_waits_have_no_hazards():
    """tools/check_lds_hazards.py: compile the MFMA kernels to ISA and replay them -- no instruction may touch a register
    whose inline-asm ds_read has not been waited for, and the MLP's prefetch AGPRs belong to the prefetch alone."""
    import shutil
    import subprocess
    import sys
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_hazards.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mlp_kernel" in r.stdout and "conv_kernel" in r.stdout and " 0 hazard(s)" in r.stdout


def test_argument_validation_without_a_gpu():
    """Unsupported shapes are rejected before any launch, with the reference's error text."""
    from scenedreamer_amd import capi
    lib = capi.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    rc = lib.sdn_grid_encode_fwd(p, p, 0, p, p, 8, 3, 3, 1, 0.0, 4, 0, p, 0, 0, None)
    assert rc == -2 and b"C must be 1, 2, 4, or 8" in lib.sdn_last_error()
    rc = lib.sdn_grid_encode_fwd(p, p, 0, p, p, 8, 6, 2, 1, 0.0, 4, 0, p, 0, 0, None)
    assert rc == -2 and b"D must be" in lib.sdn_last_error()
    rc = lib.sdn_rvip(None, p, p, p, p, p, 1.0, p, p, 2, None, p, p, p, None)
    assert rc == -1
    assert lib.sdn_posenc_fwd(None, None, 0, 5, 3, 1, None) == 0           # empty input is a no-op
    offs = (ctypes.c_int32 * 17)(*[100 * i for i in range(17)])
    rc = lib.sdn_field_collapse_table(p, ctypes.addressof(offs), 16, 0.4667, 16, p, p, None)
    assert rc == -2 and b"power-of-two" in lib.sdn_last_error()


def test_level_scales_agree_with_oracle(oracle):
    from scenedreamer_amd import capi
    S = np.float32(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    sc = np.empty(16, np.float32)
    res = np.empty(16, np.uint32)
    assert capi.lib().sdn_grid_level_scales(16, float(S), 16, sc.ctypes.data, res.ctypes.data) == 0
    for l in range(16):
        s, r = oracle.level_params(l, S, 16)
        assert sc[l] == np.float32(s) and res[l] == r


def test_ops_fail_loudly_on_cpu_tensors():
    from scenedreamer_amd import ops
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.positional_encoding(torch.zeros(4, 3), 2, -1, True)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.ray_voxel_intersection_perspective(torch.zeros(2, 2, 2, dtype=torch.int32), torch.zeros(3), torch.ones(3),
                                               torch.ones(3), 1.0, [0, 0], [2, 2], 1)
    with pytest.raises(NotImplementedError):
        ops.sp_trilinear_worldcoord()


def test_shims_are_importable():
    import scenedreamer_amd
    scenedreamer_amd.install_shims()
    import _gridencoder
    import bias_act_cuda  # noqa: F401
    import gridencoder
    import upfirdn2d_cuda  # noqa: F401
    import voxlib
    assert callable(voxlib.ray_voxel_intersection_perspective) and callable(voxlib.positional_encoding)
    assert callable(_gridencoder.grid_encode_forward) and callable(_gridencoder.grid_encode_backward)
    enc = gridencoder.GridEncoder(input_dim=5, desired_resolution=2048, level_dim=8, log2_hashmap_size=8)
    assert enc.output_dim == 128 and tuple(enc.embeddings.shape) == (16 * 256, 8)
    assert "embeddings" in enc.state_dict() and "offsets" in enc.state_dict()


def test_frame_writer_roundtrip(tmp_path):
    from scenedreamer_amd.output import FrameWriter, to_uint8_hwc
    img = torch.linspace(-1, 1, 3 * 6 * 8).reshape(1, 3, 6, 8)
    ref = ((img * 0.5 + 0.5) * 255).numpy().astype(np.uint8)[0].transpose(1, 2, 0)     # scenedreamer.py:513
    np.testing.assert_array_equal(to_uint8_hwc(img).numpy(), ref)
    w = FrameWriter(str(tmp_path), fmt="png")
    for i in range(3):
        w.submit(img, i)
    w.close()
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 3
    if files[0].endswith(".png"):
        from PIL import Image
        np.testing.assert_array_equal(np.asarray(Image.open(os.path.join(tmp_path, files[1]))), ref)
    else:
        np.testing.assert_array_equal(np.load(os.path.join(tmp_path, files[1])), ref)


def test_mp4_writer_roundtrip(tmp_path):
    """FrameWriter(video_path=...): the reference's per-frame `fout.append_data(rgb)` (scenedreamer.py:560, :631) as a
    Motion-JPEG MP4 written by scenedreamer_amd/mp4.py; box structure, frame count, fps and pixels are read back."""
    from scenedreamer_amd.mp4 import read_frames
    from scenedreamer_amd.output import FrameWriter, to_uint8_hwc
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 48), torch.linspace(-1, 1, 64), indexing="ij")
    frames = [torch.stack([torch.sin(3 * xx + k), torch.cos(2 * yy - k), xx * yy])[None] for k in range(7)]
    w = FrameWriter(str(tmp_path / "png"), fmt="png", video_path=str(tmp_path / "out.mp4"), fps=10, video_backend="mjpeg")
    for i, f in enumerate(frames):
        w.submit(f, i)
    w.close()
    assert w.frames_done == 7 and len(os.listdir(tmp_path / "png")) == 7
    fps, got = read_frames(str(tmp_path / "out.mp4"))
    assert fps == 10 and len(got) == 7
    for f, g in zip(frames, got):
        ref = to_uint8_hwc(f).numpy().astype(np.int32)
        assert g.shape == ref.shape
        assert np.abs(g.astype(np.int32) - ref).mean() < 2.0          # JPEG quality 92, 4:4:4
    raw = open(tmp_path / "out.mp4", "rb").read()
    assert raw[4:8] == b"ftyp" and b"moov" in raw and b"co64" in raw


def test_mp4_writer_strided_and_late_indices_stay_bounded(tmp_path):
    """A sharded rank submits GLOBAL frame ids (f, f + N, ...): the video thread must keep writing (bounded reorder buffer)
    instead of holding every frame until close(); a late index is still written."""
    import time

    from scenedreamer_amd.mp4 import read_frames
    from scenedreamer_amd.output import FrameWriter
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 32), torch.linspace(-1, 1, 48), indexing="ij")
    frames = [torch.stack([torch.sin(3 * xx + k), torch.cos(2 * yy - k), xx * yy])[None] for k in range(24)]
    w = FrameWriter(None, video_path=str(tmp_path / "out.mp4"), fps=10, depth=4, video_backend="mjpeg")
    order = [8 * i + 3 for i in range(20)] + [2, 500, 499, 501]
    for i, idx in enumerate(order):
        w.submit(frames[i], idx)
    deadline = time.time() + 20
    while len(w.video.sizes) < len(order) - 5 and time.time() < deadline:
        time.sleep(0.05)
    written_before_close = len(w.video.sizes)
    w.close()
    fps, got = read_frames(str(tmp_path / "out.mp4"))
    assert len(got) == len(order)
    assert written_before_close >= len(order) - 5, written_before_close     # not deferred to close()


def test_hazard_checker_finds_planted_hazards():
    """The checker itself: a rotated loop whose text order is not its execution order (the shape hipcc gives cnn.hip's
    k loop).  Clean as written; with the loop's wait weakened, or with a read landing in a fragment that is still pending,
    the control-flow walk must report it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lds_hazards", os.path.join(ROOT, "tools", "check_lds_hazards.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)

    def kernel(wait, late_dst="v[20:23]"):
        text = f"""
        ds_read_b128 v[8:11], v0 offset:0
        ds_read_b128 v[12:15], v0 offset:1024
        s_branch .LBB0_2
.LBB0_1:
        s_waitcnt lgkmcnt({wait})
        v_mfma_f32_32x32x16_f16 v[100:115], v[8:11], v[12:15], v[100:115]
        ds_read_b128 v[8:11], v0 offset:0
        ds_read_b128 v[12:15], v0 offset:1024
.LBB0_2:
        s_waitcnt lgkmcnt(2)
        ds_read_b128 {late_dst}, v0 offset:2048
        s_cbranch_scc1 .LBB0_1
        s_waitcnt lgkmcnt(0)
        v_mov_b32_e32 v1, v20
        """
        return list(enumerate(text.strip("\n").split("\n"), 1))

    n, problems = chk.check_kernel("conv_kernel_test", kernel(0))
    assert n == 5 and problems == []
    _, problems = chk.check_kernel("conv_kernel_test", kernel(2))          # the MFMA reads fragments still in flight
    assert any("touches a pending register" in why for _, _, why in problems)
    _, problems = chk.check_kernel("conv_kernel_test", kernel(0, "v[8:11]"))   # lands in a fragment whose read is in flight
    assert any("overwrites a pending destination" in why for _, _, why in problems)


def test_scene_maps_written_like_the_reference(tmp_path, scene256):
    """semantic_map.png / height_map.png (scenedreamer.py:532-545, :562-563): class colours by argmax of the one-hot map,
    height through write_img's uint8 mapping."""
    from PIL import Image
    from scenedreamer_amd.output import BIOME_COLORS, write_scene_maps
    sem, height = write_scene_maps(str(tmp_path), scene256)
    cls = torch.argmax(scene256.current_semantic_map, dim=1)[0].numpy()
    S = cls.shape[0]
    assert sem.shape == (S, S, 3) and height.shape == (S, S) and len(BIOME_COLORS) == scene256.current_semantic_map.shape[1] == 11
    want = np.asarray(BIOME_COLORS, np.int32)[cls]
    assert np.abs(sem.astype(np.int32) - want).max() <= 1            # the reference's float round trip truncates: off by <= 1
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "semantic_map.png")), sem)
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "height_map.png")), height)
    h = scene256.current_height_map[0, 0].numpy()
    np.testing.assert_array_equal(height, ((h * 0.5 + 0.5) * 255).astype(np.uint8))


@pytest.mark.needs_reference
def test_bench_cpu_baseline_reference_leg_equals_the_oracle(scene256, weights_full, lut):
    """bench.py's cpu_baseline with kind "reference" (the unmodified generator on the reference's own native sources compiled
    for the host) renders the same pixels as the oracle's tiled frame -- the two legs are interchangeable as a baseline and
    as the CPU side of the bench line's `precision` record."""
    import importlib.util
    from oracle import field_ref as FR
    from oracle import ref_native
    from scenedreamer_amd import camera, synth
    if not ref_native.available():
        pytest.skip("oracle/_ref not built")
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    z = FR.style_mlp(weights_full, np.asarray(synth.make_style(8888))).numpy()
    ge = FR.world_encoder(weights_full, scene256.current_height_map, scene256.current_semantic_map).numpy()
    pose = camera.eval_camera_poses(scene256, maxstep=40)[8]
    hw, ns, picks = (150, 200), 12, [(0, 0), (1, 1)]
    t_frame, t_tiles, got = bench._reference_tiles(weights_full, scene256, scene256.voxel_t.numpy(), pose, hw, ns, z, ge, picks)
    ref = FR.render_frame_tiled(weights_full, lut, scene256.voxel_t.numpy(), (pose[0].numpy(), pose[1].numpy(), pose[2].numpy(), pose[3]),
                                hw, ns, z, ge, tiles=picks)
    assert t_frame > 0 and t_tiles > 0 and set(got) == set(ref)
    for k in picks:
        assert got[k][:2] == ref[k][:2]
        assert torch.equal(got[k][2], ref[k][2])


def test_precision_decisions_follow_the_measurements():
    """Renderer.adopt_precision is a pure function of calibrate_style's measurements (dist.agree_precision re-runs it on the
    MAX over ranks): every gate opens and closes at its bound, explicit settings are recorded but not overridden."""
    from scenedreamer_amd import renderer as rmod
    R = object.__new__(rmod.Renderer)

    def meas(**kw):
        m = dict(field_err={6: 6e-4, 3: 6e-4}, colour_diff=5e-5, image_err={1: 4e-4, 3: 2e-4}, cnn_diff=4e-4, sky_err={3: 4e-6, 6: 1e-4},
                 explicit_colour=None, explicit_cnn=None, explicit_sky=None, pixels=518400, rays=564300, samples_per_ray=24, frame="t")
        m.update(kw)
        return m
    g = R.adopt_precision(meas())
    assert g["path"] == "fused" and g["colour"]["terms"] == 6 and R.colour_terms_auto == 6 and R.sky_terms_auto == 6
    assert R.cnn_calibration["terms3x3"] == 1 and g["sky"]["hidden_terms"] == 6 and g["image_err_vs_fp32"] == 4e-4
    g = R.adopt_precision(meas(colour_diff=2e-4, sky_err={3: 4e-6, 6: 3e-4}))
    assert g["colour"]["terms"] == 3 and R.colour_terms_auto == 3 and R.sky_terms_auto is None and g["sky"]["hidden_terms"] == 3
    g = R.adopt_precision(meas(cnn_diff=6e-4))                                   # 1-term too far from 3-term
    assert R.cnn_calibration["terms3x3"] == 3 and g["path"] == "fused" and g["image_err_vs_fp32"] == 2e-4
    g = R.adopt_precision(meas(image_err={1: 9e-4, 3: 2e-4}))                    # 1-term outside the image budget
    assert R.cnn_calibration["terms3x3"] == 3 and g["path"] == "fused"
    g = R.adopt_precision(meas(image_err={1: 3e-3, 3: 1.2e-3}, cnn_diff=2e-3))   # even the 3-term image is out: fp32 path
    assert g["path"] == "unfused" and R.field_falls_back()
    g = R.adopt_precision(meas(field_err={6: 1.2e-3, 3: 1.2e-3}))                # the field itself is out
    assert g["path"] == "unfused"
    g = R.adopt_precision(meas(field_err={6: 9.9e-4, 3: 9.9e-4}))                # ... the bound is the radiance tolerance
    assert g["path"] == "fused" and g["bound"] == rmod.FIELD_AUTO_BOUND == 1e-3
    R.cnn_calibration = None
    g = R.adopt_precision(meas(explicit_colour=3, explicit_cnn=1, explicit_sky=3, field_err={3: 5e-4}, image_err={1: 4e-4}))
    assert g["colour"] == {"terms": 3, "set_explicitly": True} and R.colour_terms_auto is None and R.cnn_calibration is None
    assert g["sky"]["set_explicitly"] and R.sky_terms_auto is None
    # the ladder between "all one product" and "all 3-term" (cnn.CNN_LADDER): the cheapest rung inside both bounds is taken
    ladder = dict(cnn_diff=6e-4, cnn_diffs={1: 6e-4, "1113": 4.5e-4, "1133": 3e-4}, image_err={1: 7e-4, "1113": 5e-4, "1133": 4e-4, 3: 2e-4})
    g = R.adopt_precision(meas(**ladder))
    assert R.cnn_calibration["terms3x3"] == "1113" and g["path"] == "fused" and g["image_err_vs_fp32"] == 5e-4
    g = R.adopt_precision(meas(**dict(ladder, cnn_diffs={1: 6e-4, "1113": 5.5e-4, "1133": 3e-4})))
    assert R.cnn_calibration["terms3x3"] == "1133" and g["image_err_vs_fp32"] == 4e-4
    g = R.adopt_precision(meas(**dict(ladder, image_err={1: 9e-4, "1113": 8.5e-4, "1133": 8.1e-4, 3: 2e-4})))
    assert R.cnn_calibration["terms3x3"] == 3 and g["path"] == "fused"
    R.cnn_auto_bound = 1e-7
    R.adopt_precision(meas())
    assert R.cnn_calibration["terms3x3"] == 3 and R.cnn_calibration["bound"] == 1e-7


def test_cnn_forms():
    from scenedreamer_amd.cnn import CNN_LADDER, form_cost, form_key, form_terms
    assert form_key(1) == 1 and form_key("1111") == 1 and form_key((3, 3, 3, 3)) == 3 and form_key([1, 1, 1, 3]) == "1113"
    assert form_terms("1133") == (1, 1, 3, 3) and form_terms(3) == (3, 3, 3, 3)
    assert [form_cost(f) for f in CNN_LADDER] == sorted(form_cost(f) for f in CNN_LADDER) and CNN_LADDER[0] == 1 and CNN_LADDER[-1] == 3
    assert form_key("1") == 1 and form_key(" 3 ") == 3 and form_key("1113") == "1113"      # SDN_CNN_TERMS arrives as a string
    with pytest.raises(ValueError, match="1113"):
        form_key("113")


def test_trunk_weight_range_is_checked_before_packing():
    """The packed trunk weights carry 2^sdn_field_trunk_shift(): a style whose weights would leave f16's range there is
    refused with a message instead of rendering infinities."""
    import torch
    from scenedreamer_amd import capi, fused
    shift = capi.lib().sdn_field_trunk_shift()
    assert 0 <= shift <= 12
    w1 = torch.full((256, 128), 0.05)
    hidden = [torch.full((256, 256), 0.1) for _ in range(3)]
    assert fused.check_trunk_range(w1, hidden, shift) == pytest.approx(0.05)
    hidden[1][3, 7] = -fused.TRUNK_F16_HEADROOM / 0.4 / 2.0 ** shift * 1.01
    with pytest.raises(RuntimeError, match="leaves f16's range"):
        fused.check_trunk_range(w1, hidden, shift)
    w1[0, 0] = float("nan")
    with pytest.raises(RuntimeError):
        fused.check_trunk_range(w1, [torch.zeros(2, 2)] * 3, shift)


def test_f16_split_loses_bits_to_subnormals_unless_the_weights_are_scaled():
    """Why pack_kernel stores the trunk weights times 2^TRUNK_SHIFT (field.hip): the lo half of the 3-term split of a weight of
    magnitude 0.03 lies in f16's subnormal range.  Emulated with numpy's float16 (which keeps subnormals, like the MFMA
    operands) on four 256 -> 256 layers with fp64 accumulation, so that only the operand representation differs."""
    rng = np.random.default_rng(0)

    def split(v, shift=0):
        v = v * 2.0 ** shift
        hi = v.astype(np.float16).astype(np.float64)
        return hi, (v - hi).astype(np.float32).astype(np.float16).astype(np.float64)

    def run(shift):
        x = rng.standard_normal((512, 256)).astype(np.float32).astype(np.float64)
        ref = x.copy()
        for _ in range(4):
            W = (rng.standard_normal((256, 256)) * 0.0625 * 0.4).astype(np.float32).astype(np.float64)
            b = rng.standard_normal(256) * 0.1
            yr = ref @ W.T + b
            ref = 1.5 * yr + np.abs(yr)
            (wh, wl), (xh, xl) = split(W, shift), split(x)
            y = (xh @ wh.T + xh @ wl.T + xl @ wh.T) * 2.0 ** -shift + b      # the three products of the kernel, descaled at the bias
            x = (1.5 * y + np.abs(y)).astype(np.float32).astype(np.float64)
        return float(np.abs(x - ref).mean() / np.abs(ref).mean())

    rng = np.random.default_rng(0); plain = run(0)
    rng = np.random.default_rng(0); scaled = run(8)
    assert plain > 4 * scaled, (plain, scaled)          # measured: 1.1e-6 vs 1.3e-7
    assert scaled < 3e-7                                 # below what fp32 operands give (4.6e-7)


def test_blocked_ray_order_host_side():
    """fused.Window: when a launch takes its rays in 8 x 4 pixel blocks (csrc/field.hip RayWindow::pix, `window_host[5]`) and how the
    host regroups per-ray data into the launch's 32-ray groups -- against the kernel's index formula restated here."""
    import torch
    from scenedreamer_amd import fused
    H0, W0, o = 102, 134, 11
    win = fused.Window.crop(H0, W0, o)
    rows, cols = H0 - 2 * o, W0 - 2 * o
    assert (rows, cols) == (80, 112) and win.blocked(0, win.n_rays) and list(win.host(0, win.n_rays))[5] == 1
    assert not win.blocked(32, win.n_rays - 32) and list(win.host(32, win.n_rays - 32))[5] == 0          # a chunk: row-major
    assert not fused.Window.crop(H0, W0, 10).blocked(0)                                                  # 82 x 114: not whole blocks
    assert not fused.Window(1000).blocked(0) and list(fused.Window(1000).host())[3:] == [0, 0, 0]        # no window at all
    bx = cols // 8

    def pix(r):          # RayWindow::pix, tiled_bx = cols / 8, ray0 = 0
        b, within = r >> 5, r & 31
        by, bxi = divmod(b, bx)
        return (4 * by + (within >> 3)) * cols + 8 * bxi + (within & 7)
    per_ray = torch.arange(win.n_rays)
    g = win.groups(per_ray)
    assert tuple(g.shape) == (win.n_rays // 32, 32)
    want = torch.tensor([pix(r) for r in range(win.n_rays)]).view(-1, 32)
    assert torch.equal(g, want) and sorted(g.reshape(-1).tolist()) == list(range(win.n_rays))         # a permutation of the window's pixels
    # every group is an 8-wide, 4-high block of the window
    y, x = g // cols, g % cols
    assert bool(((y.max(1).values - y.min(1).values) == 3).all()) and bool(((x.max(1).values - x.min(1).values) == 7).all())
    # source rays: first + y * pitch + x
    src = win.first + y * win.pitch + x
    assert int(src.min()) == o * W0 + o and int(src.max()) == (H0 - o - 1) * W0 + (W0 - o - 1)
    os.environ["SDN_RAY_BLOCKS"] = "0"
    try:
        assert not win.blocked(0, win.n_rays) and torch.equal(win.groups(per_ray), per_ray.view(-1, 32))
    finally:
        del os.environ["SDN_RAY_BLOCKS"]
    # balanced bands fall on multiples of 4 rows, so a band (+ 2 x 4 apron rows, width + 8) is whole blocks too
    from scenedreamer_amd import dist as sdist
    costs = np.r_[np.full(700, 0.2), np.linspace(0.2, 3.0, 800), np.full(660, 3.0)]
    bands = sdist.balanced_row_bands(costs, 8)
    assert all(b[0] % 4 == 0 for b in bands) and bands[-1][1] == 2160
    assert all(fused.Window((b[1] - b[0] + 30) * 3870, 3870, 11 * 3870 + 11, b[1] - b[0] + 8, 3848).blocked(0) for b in bands)


def test_ragged_blocked_ray_order_host_side():
    """A whole window of ANY size (the reference's 570 x 990 padded frame: 990 = 8 * 123 + 6) takes the blocked order in
    sdn_field_render (`window_host[5] == 2`): the block grid covers the window, positions outside it are no rays.  The host's
    regrouping against the kernel's index formula (RayWindow::pix with `rows` > 0), the group count, and that only
    field_render's callers opt in (the two-kernel launches keep whole blocks only)."""
    import torch
    from scenedreamer_amd import fused
    for rows, cols in ((82, 114), (570, 990), (5, 9), (4, 8)):
        win = fused.Window(rows * cols + 17, cols, 3, rows, cols)
        whole = cols % 8 == 0 and rows % 4 == 0
        assert win.blocked(0, win.n_rays) == whole and win.blocked(0, win.n_rays, ragged=True)
        assert list(win.host(0, win.n_rays))[5] == (1 if whole else 0) and list(win.host(0, win.n_rays, True))[5] == (1 if whole else 2)
        bx, by = -(-cols // 8), -(-rows // 4)
        assert win.n_groups(True) == bx * by >= (win.n_rays + 31) // 32 and win.n_groups() == ((bx * by) if whole else (win.n_rays + 31) // 32)

        def pix(r):          # RayWindow::pix, tiled_bx = ceil(cols / 8), rows > 0
            b, within = r >> 5, r & 31
            byi, bxi = divmod(b, bx)
            y, x = 4 * byi + (within >> 3), 8 * bxi + (within & 7)
            return y * cols + x if (x < cols and y < rows) else -1
        g = win.groups(torch.arange(win.n_rays) + 1, ragged=True) - 1          # (-1: a block position outside the window)
        want = torch.tensor([pix(r) for r in range(32 * bx * by)]).view(-1, 32)
        assert torch.equal(g, want)
        inside = g[g >= 0]
        assert sorted(inside.tolist()) == list(range(win.n_rays))               # every pixel of the window exactly once
    assert not fused.Window(1000).blocked(0, None, True)                        # no window structure: nothing to block


def test_weight_ring_protocol_with_restarts():
    """A model of field_kernel's LDS weight ring (csrc/field.hip: Ring, ring_acquire, ring_restart): 4 positions, the DMA runs 3 slots
    ahead, a pass streams 46 slots (fc_1: 4, fc_2..fc_6: 8 each, fc_out_c: 2) -- or only the first 28 when its colour branch is
    skipped, after which ring_restart refills the three positions in flight with the NEXT pass's first slots.  Invariants checked over
    random skip patterns: every acquire finds the slot it expects in its position; a refill never lands on the position of the slot
    that is current or of the one consumed just before (which slower waves may still be reading); the restart leaves exactly the
    state the kernel starts in."""
    import random
    NSLOT, AHEAD, PER_PASS, TRUNK = 4, 3, 46, 28
    rng = random.Random(5)
    for trial in range(20):
        content = [None] * NSLOT          # (pass, slot in pass) whose weights a position holds
        g, next_in_pass, issue_pass = 0, AHEAD, 0
        for sl in range(AHEAD):
            content[sl] = (0, sl)
        last_consumed_pos = None
        consumed = set()

        def acquire(expect):
            nonlocal g, next_in_pass, issue_pass, last_consumed_pos
            pos = g % NSLOT
            assert content[pos] == expect, (trial, expect, content, g)
            dst = (g + AHEAD) % NSLOT                                   # refill of slot g + 3 goes where slot g - 1 was
            assert dst == (g - 1) % NSLOT and dst != pos
            assert content[dst] is None or content[dst] in consumed, (trial, content[dst])   # never over weights still to be used
            consumed.add(expect)
            content[dst] = (issue_pass, next_in_pass)
            next_in_pass += 1
            if next_in_pass == PER_PASS:
                next_in_pass, issue_pass = 0, issue_pass + 1
            last_consumed_pos = pos
            g += 1

        def restart(next_pass):
            nonlocal next_in_pass, issue_pass
            for sl in range(AHEAD):
                dst = (g + sl) % NSLOT
                assert dst != last_consumed_pos                        # the last slot of fc_4 may still be read by a slower wave
                assert content[dst] is None or content[dst] in consumed or content[dst][0] == next_pass - 1   # stale slots of the skipped colour branch
                content[dst] = (next_pass, sl)
            next_in_pass, issue_pass = AHEAD, next_pass

        for p in range(200):
            skip = rng.random() < (0.0, 0.3, 0.6, 1.0)[trial % 4]
            for sl in range(TRUNK if skip else PER_PASS):
                acquire((p, sl))
            if skip:
                restart(p + 1)
                # exactly the state a kernel start has, shifted by g: the next three positions hold slots 0, 1, 2 of the next pass
                assert [content[(g + sl) % NSLOT] for sl in range(AHEAD)] == [(p + 1, sl) for sl in range(AHEAD)] and next_in_pass == AHEAD


# ---- bench.py's printed line (a 22.6 KB line the driver could not parse = an unmeasured round) -----------------
def _canned_bench_record():
    """The last full record a builder run produced in the old format (22.6 KB on one line): the worst case in size."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_bench_final.json")
    return json.loads(open(path).read().strip().splitlines()[-1])


def test_bench_line_is_small_numeric_and_complete():
    import json
    from scenedreamer_amd import benchline
    full = _canned_bench_record()
    assert len(json.dumps(full)) > 20000                 # the record the driver could not read
    # the records this round adds, as bench.py fills them
    full["floor"] = {"frames_per_s": 41.234567, "max_abs_err_vs_fp32": 4.4e-4, "weights": "x" * 300}
    full["colour_skip_off_frames_per_s"] = 50.123456
    full["style_cost"] = {"style_setup_ms": 3.21, "calibration_ms": 401.5, "first_frame_ms": 433.0, "trajectory40_frames_per_s": 36.6, "adopted": {"cnn": 1}}
    s = benchline.line(full, "/somewhere/bench_detail.json")
    assert "\n" not in s and len(s.encode()) < benchline.LINE_LIMIT <= 6144
    rec = json.loads(s)
    assert json.loads(benchline.dumps(rec)) == rec       # round trip
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "roofline_grid_sampler", "roofline_cnn", "roofline_rvip", "roofline_sky", "cpu_baseline", "precision"):
        assert k in rec, k
    assert rec["metric"] == full["metric"] and rec["steps"] == full["steps"] and rec["warmup"] == full["warmup"]
    assert abs(rec["value"] - full["value"]) < 1e-3 * full["value"] and abs(rec["ms_per_step"] - full["ms_per_step"]) < 1e-3 * full["ms_per_step"]
    assert set(rec["config"]) >= {"workload", "baseline_config", "path", "apron"} and "model" not in rec["config"]
    roof = rec["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_finished_samples", "avg_launch_ms", "samples_evaluated", "traffic"):
        assert k in roof, k
    assert len(roof["kernel"]) <= 80 and roof["bound"] == "mfma" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert set(rec["cpu_baseline"]) >= {"value", "unit", "cores", "kind"} and rec["cpu_baseline"]["kind"] in ("reference", "port")
    assert set(rec["precision"]) >= {"max_abs_err", "bound"}
    for k in ("dropin_frames_per_s", "config3_frames_per_s", "config5_1gpu_frames_per_s", "fallback_fp32_frames_per_s", "floor_frames_per_s",
              "colour_skip_off_frames_per_s", "style_setup_ms", "calibration_ms", "first_frame_ms", "trajectory40_frames_per_s"):
        assert isinstance(rec[k], float), k

    # nothing but numbers, booleans, null and SHORT strings; no nested prose
    def walk(v, depth=0):
        if isinstance(v, dict):
            assert depth < 2
            for x in v.values():
                walk(x, depth + 1)
        elif isinstance(v, list):
            assert all(isinstance(x, (int, float)) for x in v)
        elif isinstance(v, str):
            assert len(v) <= 120, v
        else:
            assert v is None or isinstance(v, (bool, int, float))
    walk(rec)


def test_bench_line_survives_missing_and_oversized_records():
    import json
    from scenedreamer_amd import benchline
    # a multi-rank / --profile run: no cpu_baseline, no precision error, no extras; NaN must not reach the line
    rec = json.loads(benchline.line({"metric": "m", "value": float("nan"), "unit": "frames/s", "n_gpus": 8, "steps": 20, "warmup": 5,
                                     "ms_per_step": 2.5, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                                     "data": "synthetic", "config": {"workload": "w" * 500}, "roofline": None,
                                     "broadcast": {"broadcast_s": 0.031}, "other_configs": {"error": "x"}, "dropin": {"skipped": "y"}}))
    assert rec["value"] is None and rec["roofline"] is None and rec["broadcast_s"] == 0.031 and len(rec["config"]["workload"]) <= 120
    # a record that would not fit loses its optional parts, never its headline
    big = _canned_bench_record()
    big["stage_ms"] = {f"stage{i}": float(i) for i in range(600)}
    s = benchline.line(big)
    assert len(s) < benchline.LINE_LIMIT and "stage_ms" not in json.loads(s) and "roofline" in json.loads(s)


def test_stdout_guard_leaves_exactly_one_line_on_stdout():
    """Python prints, C-level writes and child processes between guard creation and emit all land on stderr."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, subprocess; sys.path.insert(0, %r)\n"
            "from scenedreamer_amd import benchline\n"
            "g = benchline.StdoutGuard()\n"
            "print('Rendering frame 1')\n"
            "os.write(1, b'library chatter\\n')\n"
            "subprocess.run(['echo', 'child chatter'])\n"
            "g.emit('{\"value\": 1}')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"value": 1}\n'
    assert "Rendering frame 1" in r.stderr and "library chatter" in r.stderr and "child chatter" in r.stderr


def test_busiest_window_equals_pooling():
    """renderer._busiest_window (summed-area table) picks the window a dense average pooling would, ragged sizes included."""
    import torch
    import torch.nn.functional as F
    from scenedreamer_amd.renderer import _busiest_window
    g0 = torch.Generator().manual_seed(3)
    for (H, W, Hc, Wc) in ((570, 990, 286, 286), (131, 77, 40, 77), (64, 64, 64, 64), (301, 415, 264, 264)):
        g = torch.rand(H, W, generator=g0)
        g[H // 3:H // 3 + 9, W // 2:W // 2 + 11] += 2.0
        box = F.avg_pool2d(g[None, None].double(), (Hc, Wc), stride=8)[0, 0]
        k = int(box.argmax())
        assert _busiest_window(g, Hc, Wc) == ((k // box.shape[1]) * 8, (k % box.shape[1]) * 8)
