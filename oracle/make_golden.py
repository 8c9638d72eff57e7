"""Generate tests/golden/* from the UNMODIFIED reference Python (build container only).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Runs imaginaire.generators.scenedreamer.Generator (_forward_perpix, _forward_global, style_net,
world_encoder, sky_net, EvalCameraController) on CPU with the three native ops served by the C
oracle (`--native ref`: by the reference's own sources compiled for the host, oracle/_ref -- the
two produce identical files, tests/test_ref_pin_cpu.py), on seeded synthetic inputs (scenedreamer_amd/synth.py), and records small input/output
vectors.  Also exports the minecraft-id -> reduced-label LUT (data derived from the reference's
CSV tables) to scenedreamer_amd/data/mc2reduced.json.
"""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from oracle import ref_harness as RH  # noqa: E402
from scenedreamer_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SCENE_S, SCENE_SEED, W_SEED, Z_SEED = 256, 3407, 0, 8888


def main():
    assert RH.available(), "reference tree not present"
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RH.install("ref" if "ref" in sys.argv[1:] else "oracle")
    scene = synth.make_scene(SCENE_S, SCENE_SEED)
    weights = synth.make_weights(W_SEED)
    G, cfg = RH.build_generator(weights, scene)

    # ---- label LUT (data) -------------------------------------------------------------------
    lut = G.label_trans.mcid2rdid_lut.tolist()
    with open(os.path.join(ROOT, "scenedreamer_amd", "data", "mc2reduced.json"), "w") as f:
        json.dump({"source": "imaginaire/model_utils/gancraft/mc_reduction.csv + reduced_coco_lbls.csv via "
                             "mc_lbl_reduction.py:36-43", "ignore_id": int(G.label_trans.ignore_id),
                   "dirt_id": int(G.label_trans.dirt_id), "num_reduced": int(G.num_reduced_labels), "lut": lut}, f)

    # ---- camera poses (camctl.py) -------------------------------------------------------------
    import imaginaire.model_utils.gancraft.camctl as camctl
    ctl = camctl.EvalCameraController(G.voxel, maxstep=8, pattern=0, cam_ang=72, smooth_decay_multiplier=150 / 8)
    poses = [(np.asarray(o, np.float32), np.asarray(d, np.float32), np.asarray(u, np.float32), float(f))
             for (o, d, u, f) in ctl]
    np.savez_compressed(os.path.join(GOLD, "camera_pattern0.npz"), scene_S=SCENE_S, scene_seed=SCENE_SEED, maxstep=8,
                        ori=np.stack([p[0] for p in poses]), dir=np.stack([p[1] for p in poses]),
                        up=np.stack([p[2] for p in poses]), f=np.asarray([p[3] for p in poses], np.float64))
    allp = {}
    for pat in range(10):   # all ten EvalCameraController trajectories
        c = camctl.EvalCameraController(G.voxel, maxstep=8, pattern=pat, cam_ang=72, smooth_decay_multiplier=150 / 8)
        allp[f"ori{pat}"] = np.stack([np.asarray(q[0], np.float32) for q in c])
        allp[f"dir{pat}"] = np.stack([np.asarray(q[1], np.float32) for q in c])
        allp[f"up{pat}"] = np.stack([np.asarray(q[2], np.float32) for q in c])
        allp[f"f{pat}"] = np.asarray([float(q[3]) for q in c], np.float64)
    np.savez_compressed(os.path.join(GOLD, "camera_patterns.npz"), scene_S=SCENE_S, scene_seed=SCENE_SEED, maxstep=8, **allp)

    # ---- per-trajectory constants ---------------------------------------------------------------
    style = torch.from_numpy(synth.make_style(Z_SEED))
    with torch.no_grad():
        z = G.style_net(style)
        global_enc = G.world_encoder(G.voxel.current_height_map, G.voxel.current_semantic_map)
    np.savez_compressed(os.path.join(GOLD, "style_globalenc.npz"), scene_S=SCENE_S, scene_seed=SCENE_SEED,
                        w_seed=W_SEED, z_seed=Z_SEED, z=z.numpy(), global_enc=global_enc.numpy())

    # ---- per-pixel field + CNN on small frames ----------------------------------------------------
    for tag, pose_i, hw, ns in (("a", 1, (18, 34), 24), ("b", 3, (10, 18), 12), ("c", 6, (10, 10), 40)):
        RH.set_inference_overrides(G, ns, list(hw))
        cam_ori, cam_dir, cam_up, cam_f = ctl[pose_i]
        f = cam_f * (hw[1] - 1)
        c = [(G.cam_res[0] - 1) / 2, (G.cam_res[1] - 1) / 2]
        import voxlib
        with torch.no_grad():
            vid, d2, rd = voxlib.ray_voxel_intersection_perspective(G.voxel.voxel_t, cam_ori, cam_dir, cam_up, f, c,
                                                                    G.cam_res, 6)
            vid, d2, rd = vid.unsqueeze(0), d2.unsqueeze(0), rd.unsqueeze(0)
            cam_ori_t = cam_ori.unsqueeze(0)
            # sky pre-pass, scenedreamer.py:592-598
            sky_in = voxlib.positional_encoding(rd.expand(-1, -1, -1, 1, -1).contiguous(), G.pe_params_sky[0], -1,
                                                G.pe_params_sky[1])
            G.sky_avg = torch.mean(G.sky_net(sky_in, z), dim=[1, 2], keepdim=True)
            out = G._forward_perpix(None, vid, d2.clone(), rd, cam_ori_t, z, global_enc)
            net_out, new_dists, weights_, tw, rand_depth, net_s, net_c, sky_c, nosky, sky_mask, sky_only, new_idx = out
            img, _ = G._forward_global(net_out, z)
        np.savez_compressed(
            os.path.join(GOLD, f"field_{tag}.npz"), scene_S=SCENE_S, scene_seed=SCENE_SEED, w_seed=W_SEED,
            z_seed=Z_SEED, pose_index=pose_i, maxstep=8, resolution_hw=np.asarray(hw), num_samples=ns,
            cam_ori=np.asarray(cam_ori, np.float32), cam_dir=np.asarray(cam_dir, np.float32),
            cam_up=np.asarray(cam_up, np.float32), cam_f=np.float64(f), cam_c=np.asarray(c, np.float64),
            voxel_id=vid.numpy(), depth2=d2.numpy(), raydirs=rd.numpy(), sky_avg=G.sky_avg.numpy(),
            net_out=net_out.numpy(), image=img.numpy(), rand_depth=rand_depth.numpy(),
            new_idx=new_idx.numpy().astype(np.int8), total_weights=tw.numpy(), sigma=net_s.numpy().astype(np.float32),
            color_l2=np.sqrt((net_c.numpy() ** 2).sum(-1)), z=z.numpy(), global_enc=global_enc.numpy())
        del G.sky_avg
        print(tag, "net_out", net_out.shape, float(net_out.abs().max()), "hits", float((vid[..., 0, 0] != 0).float().mean()),
              "sigma range", float(net_s.min()), float(net_s.max()), "T", float(tw.mean()))
    print("golden written to", GOLD)


if __name__ == "__main__":
    main()
