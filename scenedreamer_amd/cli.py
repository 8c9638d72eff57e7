"""Trajectory renderer with the reference's inference.py / inference_givenstyle interface
(inference.py:18-32, configs/scenedreamer_inference.yaml:1-17):

    python -m scenedreamer_amd.cli --output_dir out --seed 8888 [--checkpoint scenedreamer_released.pt]
        [--camera_mode 4 --cam_maxstep 40 --resolution_hw 540 960 --num_samples 40 --cam_ang 72 --scene_size 2048]
        [--scene world.npz]     # a saved reference world: voxel_t, heightmap, current_height_map, current_semantic_map, trans_mat
    python -m torch.distributed.run --nproc-per-node N ... -m scenedreamer_amd.cli ...     # frames sharded over GPUs

Without --checkpoint (none is available offline) a seeded synthetic scene and random-init weights of the reference's
shapes are used.  Outputs as the reference writes them (scenedreamer.py:557-564, :629-632): <output_dir>/rgb_render/%05d.png,
semantic_map.png, height_map.png, style.npy and, on a single rank, <output_dir>/rgb_render.mp4 at 10 fps -- all by a
background writer."""
import argparse
import os

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--output_dir", required=True)
    ap.add_argument("--seed", type=int, default=8888)
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--camera_mode", type=int, default=4)
    ap.add_argument("--cam_maxstep", type=int, default=40)
    ap.add_argument("--resolution_hw", type=int, nargs=2, default=[540, 960])
    ap.add_argument("--num_samples", type=int, default=40)
    ap.add_argument("--cam_ang", type=float, default=72)
    ap.add_argument("--scene_size", type=int, default=2048)
    ap.add_argument("--scene", default="", help="npz with the fields of the reference's voxel handle (pcg_gen.py:161-174)")
    ap.add_argument("--mode", default="fused", choices=["fused", "unfused"])
    args = ap.parse_args()

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="env://")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from . import camera, synth
    from . import dist as sdist
    from .output import FrameWriter, write_scene_maps
    from .renderer import Renderer

    scene = weights = style = None
    if rank == 0:
        if args.scene:
            z = np.load(args.scene)
            scene = synth.Scene()
            scene.voxel_t = torch.from_numpy(z["voxel_t"].astype(np.int32)).to(dev)
            scene.heightmap = torch.from_numpy(z["heightmap"])
            scene.current_height_map = torch.from_numpy(z["current_height_map"].astype(np.float32)).to(dev)
            scene.current_semantic_map = torch.from_numpy(z["current_semantic_map"].astype(np.float32)).to(dev)
            scene.trans_mat = torch.from_numpy(z["trans_mat"].astype(np.float32))
            scene.sample_size = int(scene.voxel_t.shape[1])
        else:
            scene = synth.make_scene(args.scene_size, 3407, device=dev)
        if args.checkpoint:
            ck = torch.load(args.checkpoint, map_location="cpu")      # inference.py:57-61 ("module." prefix)
            weights = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ck["net_G"].items()}
        else:
            weights = synth.make_weights(0)
        style = synth.make_style(args.seed)                           # inference.py:81-82
    if world > 1:
        scene, weights, style = sdist.broadcast_state(scene, weights, style, dev, src=0)
    R = Renderer(weights, scene, dev)
    R.set_style(style)
    poses = camera.eval_camera_poses(scene, maxstep=args.cam_maxstep, pattern=args.camera_mode, cam_ang=args.cam_ang)
    if world > 1 and args.mode == "fused":     # one decision of the render CNN's precision gate for all ranks
        sdist.agree_cnn_precision(R, poses[0], tuple(args.resolution_hw), args.num_samples)
    out = os.path.join(args.output_dir, "rgb_render")
    # (a sharded job writes PNGs from every rank; the video needs the frames in one place, so only a single rank writes it)
    writer = FrameWriter(out, video_path=(out + ".mp4") if world == 1 else None, fps=10)
    if rank == 0:
        write_scene_maps(out, scene)                                  # scenedreamer.py:562-563
        np.save(os.path.join(out, "style.npy"), np.asarray(style))    # scenedreamer.py:564
    mine = sdist.shard_frames(range(len(poses)), rank, world)
    hw = tuple(args.resolution_hw)
    if args.mode == "fused":     # next frame's ray casting on a second stream beside the current frame
        frames = R.render_frames([poses[f] for f in mine], hw, args.num_samples, mode="fused")
    else:
        frames = (R.render_frame(poses[f], hw, args.num_samples, mode=args.mode) for f in mine)
    for f, img in zip(mine, frames):
        print(f"[rank {rank}] Rendering frame {f}", flush=True)
        writer.submit(img, f)
    writer.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
