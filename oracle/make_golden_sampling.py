"""tests/golden/stochastic_sampling.npz from the UNMODIFIED reference (build container only):
imaginaire.model_utils.gancraft.mc_utils.sample_depth_batched in its training configuration (deterministic=False,
use_box_boundaries=False, scenedreamer_train.yaml:120-121) on ray-marcher output of the synthetic scene.  The uniform
randoms the reference draws (torch.rand at mc_utils.py:121) are re-drawn from the same seed and stored, so the GPU
test can feed them to sdn_sample_depth / the fused encode.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_sampling
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from oracle import ref_harness as RH  # noqa: E402
from scenedreamer_amd import camera, synth  # noqa: E402


def main():
    RH.install("oracle")
    from imaginaire.model_utils.gancraft import mc_utils
    scene = synth.make_scene(256, 3407)
    pose = camera.eval_camera_poses(scene, maxstep=8)[3]
    hw = (40, 56)
    f, c, cam_res = camera.frame_intrinsics(pose[3], hw, 0)
    _, d2, _ = O.rvip(scene.voxel_t.numpy(), pose[0].numpy(), pose[1].numpy(), pose[2].numpy(), f, c, cam_res, 6)
    depth2 = torch.from_numpy(d2)[None]                      # [1,2,H,W,M,1]
    out = {}
    for ns in (13, 25):
        torch.manual_seed(1234 + ns)
        rd, nd, idx = mc_utils.sample_depth_batched(depth2.clone(), ns, deterministic=False, use_box_boundaries=False,
                                                    sample_depth=3)
        torch.manual_seed(1234 + ns)
        u = torch.rand([1, hw[0], hw[1], ns, 1], dtype=torch.float32)     # the draw of mc_utils.py:121
        out[f"u{ns}"], out[f"depth{ns}"], out[f"dists{ns}"], out[f"idx{ns}"] = u.numpy(), rd.numpy(), nd.numpy(), idx.numpy().astype(np.int8)
        rd2, nd2, idx2 = mc_utils.sample_depth_batched(depth2.clone(), ns, deterministic=True, use_box_boundaries=False,
                                                       sample_depth=3)
        out[f"det_depth{ns}"], out[f"det_dists{ns}"], out[f"det_idx{ns}"] = rd2.numpy(), nd2.numpy(), idx2.numpy().astype(np.int8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stochastic_sampling.npz"), depth2=depth2.numpy(), **out)
    print("written", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
