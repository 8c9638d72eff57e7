"""Time the drop-in hash-grid forward (GridEncoder.forward's native call) on SceneDreamer's grid: SDN_GRID_QUAD=0|1."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenedreamer_amd import ops, synth
from scenedreamer_amd.renderer import _time_ms
w = synth.make_weights(0)
emb = w["hash_encoder.embeddings"].cuda() if isinstance(w["hash_encoder.embeddings"], torch.Tensor) else torch.as_tensor(np.asarray(w["hash_encoder.embeddings"])).cuda()
offs = torch.as_tensor(np.asarray(w["hash_encoder.offsets"])).cuda()
B = 65536 * 24
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand(B, 5, device="cuda", generator=g)
x[:, 3:] = torch.tensor([0.31, 0.77], device="cuda")          # global_enc is constant per scene
L = offs.numel() - 1
S = float(np.log2(np.exp2(np.log2(2048 / 16) / (L - 1))))
out = torch.empty(L, B, 8, device="cuda")
dummy = torch.empty(1, device="cuda")
ms = _time_ms(lambda: ops.grid_encode_forward(x, emb, offs, out, B, 5, 8, L, S, 16, False, dummy, 0, False), 10)
print(f"SDN_GRID_QUAD={os.environ.get('SDN_GRID_QUAD', '1')}: {ms:.3f} ms for {B} samples -> {B * 16916 / ms / 1e6:.0f} GB/s effective (16 916 B/sample), checksum {float(out.double().sum()):.6f}")
