"""``MSDeformAttn`` -- the caller of the op, interface- and checkpoint-compatible with the
reference's module (mask2former/modeling/pixel_decoder/ops/modules/ms_deform_attn.py:34-125):
same constructor, same four ``nn.Linear`` sub-modules (so the same state-dict keys), same
initialisation, same forward arguments and maths.  The four linears and the softmax stay on
stock PyTorch (hipBLASLt); only the sampling op is ours.

Difference by design: the reference wraps the op call in a bare ``except`` and silently
falls back to PyTorch on *any* error; here CUDA tensors always take the HIP kernels (errors
propagate) and CPU tensors take the PyTorch path by explicit device dispatch."""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from . import ext
from .functions import MSDeformAttnFunction, MSDeformAttnFusedFunction, ms_deform_attn_core_pytorch


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a per-head dimension of 32 takes the fast HIP kernels; "
                          "other sizes run the generic ones.")
        self.im2col_step = 128                      # ms_deform_attn.py:52
        # opt-in (not reference behaviour): hand the raw offsets / logits to the fused entry points
        # (softmax and location arithmetic inside the HIP kernels) when the call qualifies
        self.fused_prologue = False
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """ms_deform_attn.py:66-80: zero offset weights, offsets biased to n_heads directions of
        unit max-norm scaled by the point index, uniform attention, Xavier projections."""
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = grid / grid.abs().max(-1, keepdim=True)[0]
        grid = grid.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        grid = grid * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.reshape(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        """query [N,Lq,C]; reference_points [N,Lq,L,2] in [0,1] (or [N,Lq,L,4] boxes);
        input_flatten [N,S,C]; input_spatial_shapes [L,2] (H,W); input_level_start_index [L];
        input_padding_mask [N,S] True = padding.  Returns [N,Lq,C].  (ms_deform_attn.py:82-125)"""
        n, len_q, _ = query.shape
        _, len_in, _ = input_flatten.shape
        assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == len_in

        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(n, len_in, self.n_heads, self.d_model // self.n_heads)
        offsets = self.sampling_offsets(query).view(n, len_q, self.n_heads, self.n_levels, self.n_points, 2)
        weights = self.attention_weights(query).view(n, len_q, self.n_heads, self.n_levels * self.n_points)
        if self.fused_prologue and ext.fused_forward_supported(value, reference_points, self.n_levels,
                                                               self.n_points, offsets, weights):
            output = MSDeformAttnFusedFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                                     reference_points.contiguous(), offsets, weights)
            return self.output_proj(output)
        weights = F.softmax(weights, -1).view(n, len_q, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            locations = reference_points[:, :, None, :, None, :] \
                + offsets / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead."
                             .format(reference_points.shape[-1]))
        if value.is_cuda:
            output = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                                locations, weights, self.im2col_step)
        else:
            output = ms_deform_attn_core_pytorch(value, input_spatial_shapes, locations, weights)
        return self.output_proj(output)
