"""CPU ORACLE for the Python layers of the hot path (TEST INFRASTRUCTURE ONLY).

A literal torch-CPU fp32 restatement of the reference's per-pixel renderer
(sample placement -> hash grid -> style-modulated MLP -> volume rendering ->
sky compositing -> render CNN), one function per reference function, each citing
the file:line it follows (paths relative to /root/reference).  The three native
ops it needs come from the C oracle (oracle/sdn_oracle.c).

Pinned against the real thing: tests/golden/field_*.npz were produced by
oracle/make_golden.py, which imports the UNMODIFIED reference generator
(imaginaire.generators.scenedreamer.Generator._forward_perpix / _forward_global)
in the build container and records its outputs; tests/test_oracle_golden.py
checks this restatement against those fixtures.

`dtype=torch.float64` evaluates the same graph in double for error-budget
studies (the native ops stay fp32).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O


def T(w, name, dtype=torch.float32):
    return torch.as_tensor(np.asarray(w[name]), dtype=dtype)


# --------------------------------------------------------------------------- a-13 once per trajectory

def style_mlp(w, z, dtype=torch.float32):
    """StyleMLP.forward, gancraft_base.py:113-126 (normalize_input=True, output_act=True)."""
    z = torch.as_tensor(z, dtype=dtype)
    z = F.normalize(z, p=2, dim=-1)
    for i in range(5):
        z = F.leaky_relu(F.linear(z, T(w, f"style_net.fc_layers.{i}.weight", dtype), T(w, f"style_net.fc_layers.{i}.bias", dtype)), 0.2)
    z = F.linear(z, T(w, "style_net.fc_out.weight", dtype), T(w, "style_net.fc_out.bias", dtype))
    return F.leaky_relu(z, 0.2)


def world_encoder(w, height_map, semantic_map, dtype=torch.float32):
    """ConditionalHashGrid.forward, layers.py:40-55."""
    act = lambda x: F.leaky_relu(x, 0.2)
    hm = torch.as_tensor(height_map, dtype=dtype)
    sm = torch.as_tensor(semantic_map, dtype=dtype)
    h = act(F.conv2d(hm, T(w, "world_encoder.hconv_head.weight", dtype), T(w, "world_encoder.hconv_head.bias", dtype), stride=2, padding=1))
    s = act(F.conv2d(sm, T(w, "world_encoder.sconv_head.weight", dtype), T(w, "world_encoder.sconv_head.bias", dtype), stride=2, padding=1))
    joint = torch.cat([h, s], dim=1)
    for i in range(5):
        # SRTConvBlock, layers.py:6-23: conv-relu-conv(stride 2)-relu, then the outer LeakyReLU
        x = F.relu(F.conv2d(joint, T(w, f"world_encoder.conv_blocks.{i}.layers.0.weight", dtype), None, stride=1, padding=1))
        x = F.relu(F.conv2d(x, T(w, f"world_encoder.conv_blocks.{i}.layers.2.weight", dtype), None, stride=2, padding=1))
        joint = act(x)
    out = joint.permute(0, 2, 3, 1)
    out = torch.mean(out.reshape(out.shape[0], -1, out.shape[-1]), dim=1)
    cond = act(F.linear(out, T(w, "world_encoder.fc1.weight", dtype), T(w, "world_encoder.fc1.bias", dtype)))
    return torch.tanh(F.linear(cond, T(w, "world_encoder.fc2.weight", dtype), T(w, "world_encoder.fc2.bias", dtype)))


# --------------------------------------------------------------------------- a-3

def cumsum_exclusive(t, dim):
    """mc_utils.py:75-79."""
    c = torch.cumsum(t, dim)
    c = torch.roll(c, 1, dim)
    c.index_fill_(dim, torch.tensor([0], dtype=torch.long), 0)
    return c


def sample_depth_batched(depth2, nsamples, sample_depth, rand=None):
    """mc_utils.py:82-151 with use_box_boundaries=False; deterministic=True unless `rand` [bs,dim0,dim1,nsamples,1] (what
    the reference draws with torch.rand at :121) is given: then the stochastic stratified branch :121-124."""
    bs, dim0, dim1 = depth2.size(0), depth2.size(2), depth2.size(3)
    dists = depth2[:, 1] - depth2[:, 0]
    dists[torch.isnan(dists)] = 0
    accu_depth = torch.cumsum(dists, dim=-2)
    total_depth = accu_depth[..., [-1], :]
    total_depth = torch.clamp(total_depth, None, sample_depth)
    if rand is None:
        rand_samples = torch.empty([bs, dim0, dim1, nsamples, 1], dtype=total_depth.dtype)
        rand_samples[..., :, 0] = torch.linspace(0, 1, nsamples + 2)[1:-1]
    else:
        rand_samples = torch.as_tensor(rand, dtype=total_depth.dtype).clone()
        rand_samples = rand_samples / nsamples
        rand_samples[..., :, 0] += torch.linspace(0, 1, nsamples + 1)[:-1]
    rand_samples = rand_samples * total_depth
    rand_samples, _ = torch.sort(rand_samples, dim=-2, descending=False)
    midpoints = (rand_samples[..., 1:, :] + rand_samples[..., :-1, :]) / 2
    new_dists = rand_samples[..., 1:, :] - rand_samples[..., :-1, :]
    idx = torch.sum(midpoints.unsqueeze(-3) > accu_depth.unsqueeze(-2), dim=-3)
    depth_deltas = depth2[:, 0, :, :, 1:, :] - depth2[:, 1, :, :, :-1, :]
    depth_deltas = torch.cumsum(depth_deltas, dim=-2)
    depth_deltas = torch.cat([depth2[:, 0, :, :, [0], :], depth_deltas + depth2[:, 0, :, :, [0], :]], dim=-2)
    heads = torch.gather(depth_deltas, -2, idx)
    return heads + midpoints, new_dists, idx


# --------------------------------------------------------------------------- a-6 / a-2 through the C oracle

def grid_encoder(w, x, dtype=torch.float32):
    """GridEncoder.forward, gridencoder/grid.py:140-156 (+ _grid_encode.forward :22-59)."""
    x = (x + 1) / 2
    prefix = list(x.shape[:-1])
    inp = x.reshape(-1, 5).to(torch.float32).contiguous().numpy()
    offs = np.asarray(w["hash_encoder.offsets"], np.int32)
    L = offs.size - 1
    S = np.log2(np.exp2(np.log2(2048 / 16) / (L - 1)))  # grid.py:33 np.log2(per_level_scale)
    out = O.grid_encode_fwd(inp, np.asarray(w["hash_encoder.embeddings"], np.float32), offs, np.float32(S), 16)
    out = torch.from_numpy(out).permute(1, 0, 2).reshape(inp.shape[0], -1)  # grid.py:52
    return out.view(prefix + [out.shape[-1]]).to(dtype)


def positional_encoding(x, ndeg, incl_orig):
    """voxlib.positional_encoding on the last dim (positional_encoding.py:38-39)."""
    return torch.from_numpy(O.posenc_fwd(x.to(torch.float32).contiguous().numpy(), ndeg, -1, incl_orig))


# --------------------------------------------------------------------------- a-7 / a-8

def mod_linear(w, name, x, z, dtype):
    """ModLinear.forward, layers.py:241-271 (bias=False, mod_bias=True, output_mode=True)."""
    x_shape = x.shape
    x = x.reshape(x_shape[0], -1, x_shape[-1])
    z = z.reshape(z.shape[0], 1, z.shape[-1])
    lin = lambda v, W, b: torch.addmm(b.unsqueeze(0), v.reshape(-1, v.shape[-1]), W.t()).reshape(*v.shape[:-1], -1)
    alpha = lin(z, T(w, name + ".weight_alpha", dtype), T(w, name + ".bias_alpha", dtype))
    wt = T(w, name + ".weight", dtype).unsqueeze(0) * alpha
    beta = lin(z, T(w, name + ".weight_beta", dtype), T(w, name + ".bias_beta", dtype))
    x = torch.baddbmm(beta, x, wt.transpose(1, 2))
    return x.reshape(*x_shape[:-1], x.shape[-1])


def render_mlp(w, x, z, m, dtype=torch.float32):
    """LightningMLP.forward, layers.py:92-126 (use_seg=True, viewdir_dim=0)."""
    act = lambda v: F.leaky_relu(v, 0.2)
    z = z[:, None, None, None, :]
    f = F.linear(x, T(w, "render_net.fc_1.weight", dtype), T(w, "render_net.fc_1.bias", dtype))
    f = f + F.linear(m, T(w, "render_net.fc_m_a.weight", dtype))
    f = act(f)
    f = act(mod_linear(w, "render_net.fc_2", f, z, dtype))
    f = act(mod_linear(w, "render_net.fc_3", f, z, dtype))
    f = act(mod_linear(w, "render_net.fc_4", f, z, dtype))
    sigma = F.linear(f, T(w, "render_net.fc_sigma.weight", dtype), T(w, "render_net.fc_sigma.bias", dtype))
    f = act(mod_linear(w, "render_net.fc_5", f, z, dtype))
    f = act(mod_linear(w, "render_net.fc_6", f, z, dtype))
    c = F.linear(f, T(w, "render_net.fc_out_c.weight", dtype), T(w, "render_net.fc_out_c.bias", dtype))
    return sigma, c


def sky_mlp(w, x, z, dtype=torch.float32):
    """SKYMLP.forward, gancraft_base.py:150-169."""
    act = lambda v: F.leaky_relu(v, 0.2)
    zz = F.linear(z, T(w, "sky_net.fc_z_a.weight", dtype))
    while zz.dim() < x.dim():
        zz = zz.unsqueeze(1)
    y = act(F.linear(x, T(w, "sky_net.fc1.weight", dtype), T(w, "sky_net.fc1.bias", dtype)) + zz)
    for i in (2, 3, 4, 5):
        y = act(F.linear(y, T(w, f"sky_net.fc{i}.weight", dtype), T(w, f"sky_net.fc{i}.bias", dtype)))
    return F.linear(y, T(w, "sky_net.fc_out_c.weight", dtype), T(w, "sky_net.fc_out_c.bias", dtype))


# --------------------------------------------------------------------------- a-9

def volum_rendering_relu(sigma, dists, dim):
    """mc_utils.py:154-161."""
    free_energy = F.relu(sigma) * dists
    a = 1 - torch.exp(-free_energy.float())
    b = torch.exp(-cumsum_exclusive(free_energy, dim=dim))
    return a * b


# --------------------------------------------------------------------------- a-4, a-5, driver

def forward_perpix(w, lut, voxel_dims, voxel_id, depth2, raydirs, cam_ori_t, z, global_enc, num_samples,
                   sample_depth=3.0, dists_scale=0.25, sky_avg=None, dtype=torch.float32, return_aux=False):
    """Generator._forward_perpix + _forward_perpix_sub, scenedreamer.py:285-430, inference settings
    (deterministic sampling, keep_sky_out + keep_sky_out_avgpool + sky_global_avgpool, clip_feat_map=True).

    voxel_id [1,h,w,M,1] int32, depth2 [1,2,h,w,M,1], raydirs [1,h,w,1,3], cam_ori_t [1,3], z [1,256],
    global_enc [1,2]; lut = int64[680] minecraft id -> reduced label.  Returns net_out [1,h,w,64].
    """
    voxel_id = torch.as_tensor(voxel_id)
    depth2 = torch.as_tensor(depth2, dtype=torch.float32).clone()
    raydirs = torch.as_tensor(raydirs, dtype=torch.float32)
    cam_ori_t = torch.as_tensor(cam_ori_t, dtype=torch.float32)
    z = torch.as_tensor(z, dtype=dtype)
    global_enc = torch.as_tensor(global_enc, dtype=torch.float32)
    lut = torch.as_tensor(lut, dtype=torch.long)

    sky_mask = voxel_id[:, :, :, [-1], :] == 0        # :335
    sky_only_mask = voxel_id[:, :, :, [0], :] == 0    # :337
    rand_depth, new_dists, new_idx = sample_depth_batched(depth2, num_samples + 1, sample_depth)  # :346
    bad = torch.isnan(rand_depth) | torch.isinf(rand_depth)
    rand_depth[bad] = 0.0                              # :350-352
    worldcoord2 = raydirs * rand_depth + cam_ori_t[:, None, None, None, :]  # :354
    reduced = lut[voxel_id.long()]                     # mc_utils.py:241-246
    reduced[reduced == 0] = 3                          # ignore -> dirt
    mc_masks = torch.gather(reduced, -2, new_idx).long()
    onehot = torch.zeros(list(mc_masks.shape[:-1]) + [12], dtype=torch.float32)
    onehot.scatter_(-1, mc_masks, 1.0)                 # :359-363

    delim = torch.tensor([float(v) for v in voxel_dims], dtype=torch.float32)  # :298-299
    ncoord = worldcoord2 / delim * 2 - 1               # :300
    genc = global_enc[:, None, None, None, :].repeat(1, ncoord.shape[1], ncoord.shape[2], ncoord.shape[3], 1)
    ncoord = torch.cat([ncoord, genc], dim=-1)         # :301-302
    feature_in = grid_encoder(w, ncoord, dtype)        # :303
    net_out_s, net_out_c = render_mlp(w, feature_in, z, onehot.to(dtype), dtype)  # :305

    sky_in = positional_encoding(raydirs.expand(-1, -1, -1, 1, -1).contiguous(), 5, True)  # :368-369
    skynet_out_c = sky_mlp(w, sky_in.to(dtype), z, dtype)                                  # :370

    weights = volum_rendering_relu(net_out_s, new_dists.to(dtype) * dists_scale, dim=-2)   # :373
    weights = weights * torch.logical_not(sky_only_mask).to(dtype)                         # :376
    total_weights = torch.sum(weights, dim=-2, keepdim=True)
    is_gnd = (worldcoord2[..., [0]] <= 1.0).any(dim=-2, keepdim=True)                      # :380-381
    nosky_mask = torch.logical_or(torch.logical_not(sky_mask), is_gnd).to(dtype)           # :382-383
    sky_weight = 1.0 - total_weights
    if sky_avg is None:
        sky_avg = torch.mean(skynet_out_c, dim=[1, 2], keepdim=True)                       # :395
    skynet_out_c = skynet_out_c * (1.0 - nosky_mask) + torch.as_tensor(sky_avg, dtype=dtype) * nosky_mask  # :401
    rgbs = torch.clamp(net_out_c, -1, 1) + 1                                               # :407-413
    rgbs_sky = torch.clamp(skynet_out_c, -1, 1) + 1
    net_out = torch.sum(weights * rgbs, dim=-2, keepdim=True) + sky_weight * rgbs_sky
    net_out = net_out.squeeze(-2) - 1
    if return_aux:
        return net_out, dict(rand_depth=rand_depth, new_dists=new_dists, new_idx=new_idx, sigma=net_out_s,
                             color=net_out_c, sky=skynet_out_c, weights=weights, feature_in=feature_in,
                             worldcoord2=worldcoord2)
    return net_out


def sky_average(w, raydirs, z, dtype=torch.float32):
    """Full-frame sky pre-pass, scenedreamer.py:592-598.  raydirs [1,H,W,1,3] -> sky_avg [1,1,1,1,64]."""
    raydirs = torch.as_tensor(raydirs, dtype=torch.float32)
    sky_in = positional_encoding(raydirs.expand(-1, -1, -1, 1, -1).contiguous(), 5, True)
    out = sky_mlp(w, sky_in.to(dtype), torch.as_tensor(z, dtype=dtype), dtype)
    return torch.mean(out, dim=[1, 2], keepdim=True)


def render_cnn(w, net_out, z, dtype=torch.float32):
    """_forward_global + RenderCNN.forward, gancraft_base.py:588-603, :202-225.  net_out [1,h,w,64] -> [1,3,h,w]."""
    act = lambda v: F.leaky_relu(v, 0.2)
    x = torch.as_tensor(net_out, dtype=dtype).permute(0, 3, 1, 2).contiguous()
    z = torch.as_tensor(z, dtype=dtype)
    cond = F.linear(z, T(w, "denoiser.fc_z_cond.weight", dtype), T(w, "denoiser.fc_z_cond.bias", dtype))
    adapt = torch.chunk(cond, 4, dim=-1)
    mod = lambda v, a, b: v * (a[..., None, None] + 1) + b[..., None, None]
    cv = lambda v, n, p: F.conv2d(v, T(w, f"denoiser.{n}.weight", dtype),
                                  T(w, f"denoiser.{n}.bias", dtype) if f"denoiser.{n}.bias" in w else None, padding=p)
    y = act(cv(x, "conv1", 0))
    y = y + cv(act(cv(y, "conv2a", 1)), "conv2b", 1)
    y = act(mod(y, adapt[0], adapt[1]))
    y = y + cv(act(cv(y, "conv3a", 1)), "conv3b", 1)
    y = act(mod(y, adapt[2], adapt[3]))
    y = y + cv(act(cv(y, "conv4a", 0)), "conv4b", 0)
    y = act(y)
    y = cv(y, "conv4", 0)
    return torch.tanh(y)


def render_frame_tiled(w, lut, vox_np, pose, resolution_hw, num_samples, z, global_enc, pad=30, tile_size=128,
                       max_blocks=6, dtype=torch.float32, tiles=None):
    """inference_givenstyle's per-frame body, scenedreamer.py:573-628: ray casting on the padded frame,
    sky pre-pass, 128-px tiles with a 30-px apron, CNN per tile, crop and stitch.  Returns image [1,3,H,W].

    tiles=[(ih, iw), ...]: evaluate only these tiles of the reference's grid (full-size configurations, where
    the whole frame costs minutes of CPU) and return {(ih, iw): (row0, col0, image_tile [1,3,h,w])} with
    row0/col0 the tile's position in the cropped output frame; ray casting and the sky pre-pass still cover
    the whole padded frame, as in the reference."""
    cam_ori, cam_dir, cam_up, cam_f = pose
    H, W = resolution_hw
    cam_res = [H + pad, W + pad]
    f = cam_f * (W - 1)
    c = [(cam_res[0] - 1) / 2, (cam_res[1] - 1) / 2]
    vid, d2, rd = O.rvip(vox_np, np.asarray(cam_ori, np.float32), np.asarray(cam_dir, np.float32),
                         np.asarray(cam_up, np.float32), f, c, cam_res, max_blocks)
    vid, d2, rd = torch.from_numpy(vid)[None], torch.from_numpy(d2)[None], torch.from_numpy(rd)[None]
    cam_ori_t = torch.as_tensor(np.asarray(cam_ori, np.float32))[None]
    sky_avg = sky_average(w, rd, z, dtype)
    nh = (cam_res[0] - pad + tile_size - 1) // tile_size
    nw = (cam_res[1] - pad + tile_size - 1) // tile_size
    rows = []
    picked = {}
    for ih in range(nh):
        h0, h1 = ih * tile_size, min(ih * tile_size + tile_size + pad, cam_res[0])
        cols = []
        for iw in range(nw):
            w0, w1 = iw * tile_size, min(iw * tile_size + tile_size + pad, cam_res[1])
            if tiles is not None and (ih, iw) not in tiles:
                continue
            no = forward_perpix(w, lut, vox_np.shape, vid[:, h0:h1, w0:w1], d2[:, :, h0:h1, w0:w1], rd[:, h0:h1, w0:w1],
                                cam_ori_t, z, global_enc, num_samples, sky_avg=sky_avg, dtype=dtype)
            img = render_cnn(w, no, z, dtype)
            if pad != 0:
                img = img[:, :, pad // 2:-pad // 2, pad // 2:-pad // 2]
            if tiles is not None:
                picked[(ih, iw)] = (h0, w0, img)
                continue
            cols.append(img)
        if tiles is None:
            rows.append(torch.cat(cols, dim=-1))
    if tiles is not None:
        return picked
    return torch.cat(rows, dim=-2)
