"""Autograd wrapper, interface-identical to the reference's
``MSDeformAttnFunction`` (mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:32-49):

    out = MSDeformAttnFunction.apply(value, value_spatial_shapes, value_level_start_index,
                                     sampling_locations, attention_weights, im2col_step)

forward stores the five tensors for backward; backward always computes all three
gradients (the reference does not consult needs_input_grad either) and returns
``(grad_value, None, None, grad_sampling_loc, grad_attn_weight, None)``.
"""
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ext


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        output = ext.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                            sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, level_start, sampling_locations, attention_weights = ctx.saved_tensors
        grad_value, grad_sampling_loc, grad_attn_weight = ext.ms_deform_attn_backward(
            value, shapes, level_start, sampling_locations, attention_weights,
            grad_output.contiguous(), ctx.im2col_step)
        # (bf16 variant: the library returns grad_value in fp32; autograd wants value's dtype)
        return grad_value.to(value.dtype), None, None, grad_sampling_loc, grad_attn_weight, None


class MSDeformAttnFusedFunction(Function):
    """Opt-in fused boundary: the op together with the softmax and the
    sampling-location arithmetic of MSDeformAttn.forward (reference
    ops/modules/ms_deform_attn.py:102-109).

        out = MSDeformAttnFusedFunction.apply(value, spatial_shapes, level_start_index,
                                              reference_points, sampling_offsets, attn_logits)

    with reference_points [N,Lq,L,2], sampling_offsets [N,Lq,M,L,P,2] and attn_logits [N,Lq,M,L*P]
    (the raw outputs of the module's two linears).  Saves the raw tensors, not locations/weights;
    backward returns gradients for value, sampling_offsets and attn_logits (and for
    reference_points when it requires one)."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, reference_points, sampling_offsets,
                attn_logits):
        output = ext.ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index,
                                                  reference_points, sampling_offsets, attn_logits)
        ctx.save_for_backward(value, spatial_shapes, level_start_index, reference_points,
                              sampling_offsets, attn_logits)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, level_start, ref, offsets, logits = ctx.saved_tensors
        grad_value, grad_offsets, grad_logits = ext.ms_deform_attn_fused_backward(
            value, shapes, level_start, ref, offsets, logits, grad_output.contiguous())
        grad_ref = None
        if ctx.needs_input_grad[3]:
            # d loc / d ref = 1 and grad_offsets = grad_loc / (W_l, H_l)
            wh = shapes.flip(-1).to(grad_offsets.dtype)                       # (W_l, H_l)
            grad_ref = (grad_offsets * wh[None, None, None, :, None, :]).sum(dim=(2, 4))
        return grad_value.to(value.dtype), None, None, grad_ref, grad_offsets, grad_logits


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """CPU-tensor path of the reference's ops/functions module (ms_deform_attn_func.py:52-72):
    per level, bilinear ``grid_sample`` (zero padding, align_corners=False) of the level's map
    viewed as [N*M, D, H, W] at ``2*loc - 1``, then the attention-weighted sum over L*P.

    The reference reaches it for CPU inputs through the bare ``except`` of MSDeformAttn.forward
    (ops/modules/ms_deform_attn.py:116-121; BASELINE.json configs[0]).  Here it is selected by
    device, never by catching an error: GPU tensors always take the HIP kernels."""
    import torch
    import torch.nn.functional as F
    n, _, m, d = value.shape
    lq, nl, npts = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    sizes = [(int(h), int(w)) for h, w in value_spatial_shapes]
    levels = value.split([h * w for h, w in sizes], dim=1)
    grids = 2 * sampling_locations - 1
    per_level = []
    for lvl, (h, w) in enumerate(sizes):
        fmap = levels[lvl].flatten(2).transpose(1, 2).reshape(n * m, d, h, w)
        grid = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        per_level.append(F.grid_sample(fmap, grid, mode="bilinear", padding_mode="zeros",
                                       align_corners=False))
    weights = attention_weights.transpose(1, 2).reshape(n * m, 1, lq, nl * npts)
    out = (torch.stack(per_level, dim=-2).flatten(-2) * weights).sum(-1).view(n, m * d, lq)
    return out.transpose(1, 2).contiguous()
