"""Record the exact native calls the UNMODIFIED reference generator makes (build container only).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_call_transcript

/root/reference cannot travel to the GPU box, so "the unmodified imaginaire generator runs on the HIP shims"
cannot be executed there as such.  What can be shown on the GPU is the seam itself: this script runs
Generator._forward_perpix (the sky pre-pass and the full-frame ray cast included) on the CPU with the reference's
OWN native sources underneath (oracle/_ref), wraps `voxlib` / `_gridencoder` in a recorder, and stores every call
-- function name, every argument exactly as the reference passes it (tensor dtypes / shapes / devices, Python
scalars, lists, the placeholder dy_dx tensor ...) and the result -- in tests/golden/native_calls.npz.
tests/test_shim_replay_gpu.py replays the transcript through the modules scenedreamer_amd.install_shims() provides
(the ones `import voxlib`, `import _gridencoder` resolve to in a process running the reference on an MI355X) and
compares the results.  Large outputs are stored at a fixed random subset of rows.
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness as RH  # noqa: E402
from scenedreamer_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ROWS = 768       # rows of a large output that are kept


def _describe(a):
    if isinstance(a, torch.Tensor):
        return {"kind": "tensor", "dtype": str(a.dtype).replace("torch.", ""), "shape": list(a.shape),
                "stride": list(a.stride()), "device": a.device.type}
    if isinstance(a, np.ndarray):
        return {"kind": "ndarray", "dtype": str(a.dtype), "shape": list(a.shape)}
    if isinstance(a, (list, tuple)):
        return {"kind": type(a).__name__, "items": [_describe(x) for x in a]}
    return {"kind": type(a).__name__, "value": a if isinstance(a, (int, float, bool, str, type(None))) else repr(a)}


class Recorder:
    def __init__(self):
        self.calls, self.arrays = [], {}

    def wrap(self, modname, mod):
        out = types.ModuleType(modname)
        for name in dir(mod):
            fn = getattr(mod, name)
            if callable(fn) and not name.startswith("_"):
                setattr(out, name, self._wrapped(modname, name, fn))
        return out

    def _store(self, key, t, rows=None):
        a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        if rows is not None:
            a = a[rows] if a.ndim == 2 else a[:, rows]
        self.arrays[key] = a

    def _wrapped(self, modname, name, fn):
        def call(*args):
            i = len(self.calls)
            rec = {"module": modname, "fn": name, "args": [_describe(a) for a in args]}
            if name == "grid_encode_forward":       # inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc, dy_dx, gridtype, ac
                self._store(f"c{i}_inputs", args[0])
                rec["embeddings"] = "weights:hash_encoder.embeddings"
                self._store(f"c{i}_offsets", args[2])
                res = fn(*args)
                B = int(args[4])
                rows = np.sort(np.random.default_rng(i).choice(B, size=min(ROWS, B), replace=False))
                self._store(f"c{i}_rows", rows)
                self._store(f"c{i}_outputs_rows", args[3], rows)      # [L, rows, C]
            else:
                for k, a in enumerate(args):
                    if isinstance(a, torch.Tensor) and not (name == "ray_voxel_intersection_perspective" and k == 0):
                        self._store(f"c{i}_arg{k}", a)
                if name == "ray_voxel_intersection_perspective":
                    rec["in_voxel"] = "scene:voxel_t"
                res = fn(*args)
                outs = res if isinstance(res, (list, tuple)) else [res]
                for k, o in enumerate(outs):
                    self._store(f"c{i}_out{k}", o)
                rec["n_out"] = len(outs)
            self.calls.append(rec)
            return res
        return call


def main():
    assert RH.available(), "reference tree not present"
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RH.install("ref")
    rec = Recorder()
    sys.modules["voxlib"] = rec.wrap("voxlib", sys.modules["voxlib"])
    sys.modules["_gridencoder"] = rec.wrap("_gridencoder", sys.modules["_gridencoder"])
    scene = synth.make_scene(256, 3407)
    weights = synth.make_weights(0)
    G, cfg = RH.build_generator(weights, scene)
    import imaginaire.model_utils.gancraft.camctl as camctl
    import voxlib
    ctl = camctl.EvalCameraController(G.voxel, maxstep=8, pattern=0, cam_ang=72, smooth_decay_multiplier=150 / 8)
    hw, ns, pose_i = (10, 18), 12, 3                      # the golden case "b"
    RH.set_inference_overrides(G, ns, list(hw))
    cam_ori, cam_dir, cam_up, cam_f = ctl[pose_i]
    style = torch.from_numpy(synth.make_style(8888))
    with torch.no_grad():
        z = G.style_net(style)
        global_enc = G.world_encoder(G.voxel.current_height_map, G.voxel.current_semantic_map)
        # the calls of inference_givenstyle's frame body, scenedreamer.py:575-598 and _forward_perpix :285-430
        vid, d2, rd = voxlib.ray_voxel_intersection_perspective(G.voxel.voxel_t, cam_ori, cam_dir, cam_up,
                                                                cam_f * (hw[1] - 1),
                                                                [(G.cam_res[0] - 1) / 2, (G.cam_res[1] - 1) / 2],
                                                                G.cam_res, G.num_blocks_early_stop)
        vid, d2, rd = vid.unsqueeze(0), d2.unsqueeze(0), rd.unsqueeze(0)
        from imaginaire.model_utils.gancraft import voxlib as gvox
        sky_in = gvox.positional_encoding(rd.expand(-1, -1, -1, 1, -1).contiguous(), G.pe_params_sky[0], -1,
                                          G.pe_params_sky[1])
        G.sky_avg = torch.mean(G.sky_net(sky_in, z), dim=[1, 2], keepdim=True)
        out = G._forward_perpix(None, vid, d2.clone(), rd, cam_ori.unsqueeze(0), z, global_enc)
    np.savez_compressed(os.path.join(GOLD, "native_calls.npz"), meta=json.dumps(
        {"scene_S": 256, "scene_seed": 3407, "w_seed": 0, "case": "field_b", "calls": rec.calls}), **rec.arrays)
    for c in rec.calls:
        print(c["module"], c["fn"], [a.get("dtype", a.get("kind")) for a in c["args"]])
    g = np.load(os.path.join(GOLD, "field_b.npz"))
    assert np.allclose(out[0].numpy(), g["net_out"], atol=1e-6), "transcript run does not reproduce golden b"
    print("wrote", os.path.join(GOLD, "native_calls.npz"), os.path.getsize(os.path.join(GOLD, "native_calls.npz")), "bytes")


if __name__ == "__main__":
    main()
