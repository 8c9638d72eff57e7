import os, sys, torch
os.environ["SDN_MLP_DBG"] = "128"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, fused, synth
from scenedreamer_amd.renderer import Renderer
dev = torch.device("cuda:0")
scene = synth.make_scene(2048, 3407, device=dev)
R = Renderer(synth.make_weights(0), scene, dev)
R.set_style(synth.make_style(8888))
pose = camera.eval_camera_poses(scene, maxstep=40)[10]
with torch.no_grad():
    vid, d2, rd, cam_res = R.cast_rays(pose, (540, 960))
    n = cam_res[0] * cam_res[1]
    vid, d2, rd = vid.view(n, R.M), d2.view(2, n, R.M), rd.view(n, 3)
    sky_c, sky_avg = fused.sky_fused(R, rd)
    for _ in range(2):
        out = fused.field_fused(R, vid, d2, rd, torch.as_tensor(pose[0], dtype=torch.float32), sky_c, sky_avg, 24)
    torch.cuda.synchronize()
    t = out[:1024, :3].double().cpu()
    print("waves", t.shape[0], "staging cycles mean %.3e  total cycles mean %.3e  passes mean %.1f" % (t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean()))
    print("staging fraction %.3f   staging cycles per pass %.0f   total cycles per pass %.0f" % (t[:, 0].sum() / t[:, 1].sum(), t[:, 0].sum() / t[:, 2].sum(), t[:, 1].sum() / t[:, 2].sum()))
