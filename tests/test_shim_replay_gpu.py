"""The seam `imaginaire.generators.scenedreamer.Generator` -> native extension modules, on the MI355X.

tests/golden/native_calls.npz is the transcript of every native call the UNMODIFIED reference generator makes for
one frame (recorded in the build container by oracle/make_call_transcript.py with the reference's own native sources
underneath, and checked there to reproduce golden "b").  Here the very same calls -- same function names, argument
order, Python types, tensor dtypes / shapes, CPU camera tensors, the placeholder dy_dx -- go through the modules
`scenedreamer_amd.install_shims()` puts in front of the reference (`voxlib`, `_gridencoder`), and must return what
the reference's sources returned.  (/root/reference itself cannot travel to the GPU box; where it IS present next to
a GPU, test_unmodified_generator_on_hip_shims below runs the real thing.)"""
import importlib
import json
import sys

import numpy as np
import pytest
import torch

from conftest import bits, golden

pytestmark = pytest.mark.gpu


def _rebuild(desc, arr, dev):
    k = desc["kind"]
    if k == "tensor":
        t = torch.from_numpy(np.ascontiguousarray(arr))
        assert str(t.dtype).replace("torch.", "") == desc["dtype"] and list(t.shape) == desc["shape"]
        return t.to(dev) if dev else t
    if k in ("list", "tuple"):
        return [d["value"] for d in desc["items"]]
    if k == "float64":
        return np.float64(desc["value"])
    return desc["value"]


def test_reference_call_transcript_through_the_shims(scene256, weights_full):
    import scenedreamer_amd
    scenedreamer_amd.install_shims()
    for m in ("voxlib", "_gridencoder"):
        sys.modules.pop(m, None)
    mods = {"voxlib": importlib.import_module("voxlib"), "_gridencoder": importlib.import_module("_gridencoder")}
    assert "scenedreamer_amd/shims" in mods["voxlib"].__file__.replace("\\", "/")
    g = golden("native_calls.npz")
    meta = json.loads(str(g["meta"]))
    vox = scene256.voxel_t.cuda()
    emb = torch.from_numpy(np.asarray(weights_full["hash_encoder.embeddings"], np.float32)).cuda()
    seen = []
    for i, c in enumerate(meta["calls"]):
        fn = getattr(mods[c["module"]], c["fn"])
        seen.append(c["fn"])
        if c["fn"] == "ray_voxel_intersection_perspective":
            # in_voxel on the GPU, the three camera vectors as the CPU tensors camctl hands over (camctl.py:45-47)
            args = [vox] + [_rebuild(d, g.get(f"c{i}_arg{k}"), None) for k, d in enumerate(c["args"]) if k > 0]
            out = fn(*args)
            assert len(out) == c["n_out"] and all(o.is_cuda for o in out)
            np.testing.assert_array_equal(out[0].cpu().numpy(), g[f"c{i}_out0"])
            np.testing.assert_array_equal(bits(out[1].cpu().numpy()), bits(g[f"c{i}_out1"]))
            np.testing.assert_array_equal(bits(out[2].cpu().numpy()), bits(g[f"c{i}_out2"]))
        elif c["fn"] == "positional_encoding":
            args = [_rebuild(d, g.get(f"c{i}_arg{k}"), "cuda") for k, d in enumerate(c["args"])]
            out = fn(*args)
            np.testing.assert_allclose(out.cpu().numpy(), g[f"c{i}_out0"], rtol=1e-5, atol=1e-5)
        elif c["fn"] == "grid_encode_forward":
            d = c["args"]
            inputs = torch.from_numpy(g[f"c{i}_inputs"]).cuda()
            offsets = torch.from_numpy(g[f"c{i}_offsets"]).cuda()
            outputs = torch.empty(d[3]["shape"], dtype=torch.float32, device="cuda")
            dy_dx = torch.empty(d[11]["shape"], dtype=torch.float32, device="cuda")     # the reference's placeholder [1]
            scal = [None if x["kind"] == "tensor" else _rebuild(x, None, None) for x in d]
            fn(inputs, emb, offsets, outputs, scal[4], scal[5], scal[6], scal[7], scal[8], scal[9], scal[10], dy_dx,
               scal[12], scal[13])
            rows = g[f"c{i}_rows"]
            np.testing.assert_allclose(outputs[:, torch.from_numpy(rows).cuda()].cpu().numpy(), g[f"c{i}_outputs_rows"],
                                       rtol=0, atol=1e-5)
        else:
            raise AssertionError(f"transcript holds a call the shims do not know: {c['fn']}")
    assert seen.count("ray_voxel_intersection_perspective") == 1 and "grid_encode_forward" in seen


@pytest.mark.needs_reference
def test_unmodified_generator_on_hip_shims(scene256, weights_full):
    """imaginaire.generators.scenedreamer.Generator._forward_perpix / _forward_global, UNCHANGED, with voxlib and
    _gridencoder served by libsdnative (ref_harness.install("hip")), against the goldens.  Needs /root/reference AND a
    GPU in the same machine."""
    from oracle import ref_harness as RH
    RH.install("hip")
    import voxlib
    G, _ = RH.build_generator(weights_full, scene256)
    G = G.cuda()
    G.voxel.voxel_t = scene256.voxel_t.cuda()
    for tag in "abc":
        g = golden(f"field_{tag}.npz")
        hw, ns = [int(v) for v in g["resolution_hw"]], int(g["num_samples"])
        RH.set_inference_overrides(G, ns, hw)
        z, ge = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(g["global_enc"]).cuda()
        with torch.no_grad():
            vid, d2, rd = voxlib.ray_voxel_intersection_perspective(
                G.voxel.voxel_t, torch.from_numpy(g["cam_ori"]), torch.from_numpy(g["cam_dir"]),
                torch.from_numpy(g["cam_up"]), float(g["cam_f"]), [float(v) for v in g["cam_c"]], G.cam_res, 6)
            vid, d2, rd = vid.unsqueeze(0), d2.unsqueeze(0), rd.unsqueeze(0)
            sky_in = voxlib.positional_encoding(rd.expand(-1, -1, -1, 1, -1).contiguous(), G.pe_params_sky[0], -1,
                                                G.pe_params_sky[1])
            G.sky_avg = torch.mean(G.sky_net(sky_in, z), dim=[1, 2], keepdim=True)
            out = G._forward_perpix(None, vid, d2.clone(), rd, torch.from_numpy(g["cam_ori"])[None].cuda(), z, ge)
            img, _ = G._forward_global(out[0], z)
        del G.sky_avg
        np.testing.assert_array_equal(vid.cpu().numpy(), g["voxel_id"])
        np.testing.assert_allclose(out[0].cpu().numpy(), g["net_out"], rtol=0, atol=1e-3)
        np.testing.assert_allclose(img.cpu().numpy(), g["image"], rtol=0, atol=1e-3)


@pytest.mark.needs_reference
def test_unmodified_inference_givenstyle_loop_on_hip_shims(scene256, weights_full, tmp_path):
    """The reference's OWN frame loop -- Generator.inference_givenstyle (scenedreamer.py:479-632: camera controller, per-frame
    voxlib ray casting, sky pre-pass, 2 x 2 tiles through _forward_perpix / _forward_global, crop + stitch, write_img, video
    writer), UNCHANGED, with `voxlib` / `_gridencoder` served by libsdnative -- against this package's renderer on the same
    scene, weights, style and trajectory.  cv2 / imageio do not exist in this image: cv2.imwrite is served by Pillow and
    imageio.get_writer by a recorder (host-side file writing, not part of the path).  uint8 frames must agree to one level
    (both truncate: a 1e-4 float difference moves a pixel across an integer boundary now and then), the scene maps exactly."""
    import os

    from PIL import Image
    from oracle import ref_harness as RH
    from scenedreamer_amd import camera, synth
    from scenedreamer_amd.output import to_uint8_hwc, write_scene_maps
    from scenedreamer_amd.renderer import Renderer
    RH.install("hip")
    import cv2
    import imageio

    def imwrite(path, img, params=None):
        img = np.asarray(img)
        Image.fromarray(img[..., ::-1] if img.shape[-1] == 3 else img[..., 0]).save(path)      # cv2 takes BGR
        return True

    class _Rec:
        frames = []

        def append_data(self, rgb):
            _Rec.frames.append(np.asarray(rgb).copy())

        def close(self):
            pass

    cv2.__dict__["imwrite"] = imwrite
    cv2.__dict__["IMWRITE_PNG_COMPRESSION"] = 16
    imageio.__dict__["get_writer"] = lambda path, fps=10: _Rec()
    G, _ = RH.build_generator(weights_full, scene256)
    G = G.cuda()
    G.voxel.voxel_t = scene256.voxel_t.cuda()
    G.voxel.current_height_map = scene256.current_height_map.cuda()
    G.voxel.current_semantic_map = scene256.current_semantic_map.cuda()
    hw, ns, steps = [72, 104], 12, 3
    style = torch.from_numpy(np.asarray(synth.make_style(8888))).cuda()
    out = str(tmp_path / "ref")
    with torch.no_grad():
        G.inference_givenstyle(style, out, camera_mode=0, num_samples=ns, tile_size=64, resolution_hw=hw, cam_ang=72,
                               cam_maxstep=steps)
    rdir = os.path.join(out, "rgb_render")
    assert len(_Rec.frames) == steps and sorted(os.listdir(rdir)) == ["00000.png", "00001.png", "00002.png", "height_map.png",
                                                                      "semantic_map.png", "style.npy"]
    # ---- the same trajectory through this package
    R = Renderer(weights_full, scene256, "cuda")
    R.set_style(synth.make_style(8888))
    poses = camera.eval_camera_poses(scene256, maxstep=steps, pattern=0, cam_ang=72)
    worst, off = 0, 0.0
    for f, img in enumerate(R.render_frames(poses, tuple(hw), ns, mode="fused")):
        mine = to_uint8_hwc(img).cpu().numpy().astype(np.int32)
        ref = np.asarray(Image.open(os.path.join(rdir, f"{f:05d}.png"))).astype(np.int32)
        assert ref.shape == mine.shape == (hw[0], hw[1], 3) and ref.std() > 5
        np.testing.assert_array_equal(ref, _Rec.frames[f].astype(np.int32))       # the video got the PNG's pixels
        d = np.abs(ref - mine)
        worst, off = max(worst, int(d.max())), max(off, float((d > 0).mean()))
    print(f"inference_givenstyle (unmodified, HIP shims) vs scenedreamer_amd frames: max |diff| {worst} level, "
          f"{100 * off:.2f} % of the values differ")
    assert worst <= 1 and off < 0.2
    sem, height = write_scene_maps(str(tmp_path / "mine"), scene256)
    np.testing.assert_array_equal(np.asarray(Image.open(os.path.join(rdir, "semantic_map.png"))), sem)
    np.testing.assert_array_equal(np.asarray(Image.open(os.path.join(rdir, "height_map.png"))), height)
    np.testing.assert_array_equal(np.load(os.path.join(rdir, "style.npy")), np.asarray(synth.make_style(8888)))
