import numpy as np, torch, sys
sys.path.insert(0, '.')
from scenedreamer_amd import synth, ops
from oracle import oracle as O
w = synth.make_weights(0)
offs, emb = w["hash_encoder.offsets"], w["hash_encoder.embeddings"]
rng = np.random.default_rng(3)
B = 20000
x = rng.random((B, 5), dtype=np.float32)
x[:, 3] = 0.36; x[:, 4] = 0.63; x[::101, 1] = 1.5
S = np.float32(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
out = torch.empty(16, B, 8, device="cuda"); dy = torch.empty(1, device="cuda")
ops.grid_encode_forward(torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(), torch.from_numpy(offs).cuda(), out, B, 5, 8, 16, S, 16, False, dy, 0, False)
ref = O.grid_encode_fwd(x, emb, offs, S, 16)
got = out.cpu().numpy()
err = np.abs(got-ref).max(-1)
bad = np.argwhere(err > 1e-5)
print('n bad (level,sample):', len(bad))
for l, b in bad[:12]:
    sc, res = O.level_params(int(l), S, 16)
    pos = x[b].astype(np.float32) * np.float32(sc) + np.float32(0.5)
    print(l, b, 'err', err[l,b], 'x', x[b], 'scale', sc, 'pos', pos, 'frac', pos-np.floor(pos))
print('levels of bad:', np.bincount(bad[:,0], minlength=16))
