import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import camera, synth, fused
from scenedreamer_amd.renderer import Renderer
from scenedreamer_amd.cnn import MfmaCNN
dev = torch.device("cuda:0")
scene = synth.make_scene(256, 3407, device=dev)
R = Renderer(synth.make_weights(0, grid_log2_hashmap=10), scene, dev)
R.set_style(synth.make_style(8888))
pose = camera.eval_camera_poses(scene, maxstep=8)[1]
hw = (96, 80)
a = R.render_frame(pose, hw, 12, mode="fused", apron="minimal")
b = R.render_frame(pose, hw, 12, mode="fused", apron="reference")
d = (a - b).abs()
print("image max diff", float(d.max()), "nonzero", int((d > 0).sum()), "of", d.numel())
idx = (d[0].amax(0) > 0).nonzero()
print("rows", idx[:, 0].min().item(), idx[:, 0].max().item(), "cols", idx[:, 1].min().item(), idx[:, 1].max().item())
# field only
n1 = R.render_frame(pose, hw, 12, mode="fused", cnn=False)          # full apron [1,Hp,Wp,64]
Hp, Wp = n1.shape[1:3]
o = 11
x_full = n1
x_min = n1[:, o:Hp - o, o:Wp - o].contiguous()
cnn = MfmaCNN(R, int(os.environ.get("SDN_CNN_TERMS", "1")))
i_full = cnn(x_full)[:, :, 15:-15, 15:-15]
i_min = cnn(x_min)[:, :, 4:-4, 4:-4]
print("cnn-only diff", float((i_full - i_min).abs().max()))
t_full = R.render_cnn(x_full)[:, :, 15:-15, 15:-15]
t_min = R.render_cnn(x_min)[:, :, 4:-4, 4:-4]
print("torch cnn diff", float((t_full - t_min).abs().max()))
