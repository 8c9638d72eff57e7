"""Host-side mirror of the reference's `gridencoder` package (gridencoder/grid.py:19-156).

`GridEncoder` keeps the reference's constructor arguments, parameter / buffer
names (`embeddings`, `offsets`) and forward semantics, so a reference state dict
loads unchanged; the arithmetic runs in libsdnative through
ops.grid_encode_forward / grid_encode_backward.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops

GRIDTYPE_IDS = {"hash": 0, "tiled": 1}


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """Row offsets of every level's table (gridencoder/grid.py:113-125)."""
    max_params = 2 ** log2_hashmap_size
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        side = resolution if align_corners else resolution + 1
        rows = min(max_params, side ** input_dim)
        rows = int(np.ceil(rows / 8) * 8)
        offsets.append(offset)
        offset += rows
    offsets.append(offset)
    return np.asarray(offsets, dtype=np.int32)


class _GridEncode(torch.autograd.Function):
    """grid.py:19-87.  Output is [B, L*C]; the kernel itself produces [L, B, C]."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs, gridtype,
                align_corners):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = int(base_resolution)
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
        if calc_grad_inputs:
            dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype)
        else:
            dy_dx = torch.empty(1, device=inputs.device, dtype=embeddings.dtype)
        ops.grid_encode_forward(inputs, embeddings.contiguous(), offsets, outputs, B, D, C, L, S, H, calc_grad_inputs,
                                dy_dx, gridtype, align_corners)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, C, L, S, H, gridtype)
        ctx.calc_grad_inputs = calc_grad_inputs
        ctx.align_corners = align_corners
        return outputs.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype = ctx.dims
        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        if ctx.calc_grad_inputs:
            grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype)
        else:
            grad_inputs = torch.zeros(1, device=inputs.device, dtype=embeddings.dtype)
        ops.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
                                 ctx.calc_grad_inputs, dy_dx, grad_inputs, gridtype, ctx.align_corners)
        gi = grad_inputs.to(inputs.dtype) if ctx.calc_grad_inputs else None
        return gi, grad_embeddings, None, None, None, None, None, None


grid_encode = _GridEncode.apply


class GridEncoder(nn.Module):
    """Multi-resolution hash grid (gridencoder/grid.py:93-156)."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = GRIDTYPE_IDS[gridtype]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size
        offsets = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size,
                                align_corners)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        self.n_params = int(offsets[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> "
                f"{int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} "
                f"gridtype={self.gridtype} align_corners={self.align_corners}")

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)  # [-bound, bound] -> [0, 1]
        prefix = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        out = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                          inputs.requires_grad, self.gridtype_id, self.align_corners)
        return out.view(prefix + [self.output_dim])
