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
        # grad_value already has value's dtype (bf16 variant: rounded once by the tile kernel, no conversion pass)
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None


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
    """CPU-tensor path with the name and signature of the reference's
    ``ms_deform_attn_core_pytorch`` (ops/functions/ms_deform_attn_func.py:52-72), which the reference
    reaches for CPU inputs through the bare ``except`` of MSDeformAttn.forward
    (ops/modules/ms_deform_attn.py:116-121; BASELINE.json configs[0]).  Here it is selected by
    device, never by catching an error: GPU tensors always take the HIP kernels.

    Written as the op's own definition (the arithmetic of the HIP kernels, ``corner_setup`` in
    csrc/msda_common.inc) rather than through ``grid_sample``: pixel coordinates
    ``loc * (W, H) - 0.5``, the four neighbouring pixels gathered from the flattened level with
    zero weight outside it, bilinear weights times the attention weight, summed over levels and
    points.  Differentiable in value, locations and weights (gather / index arithmetic only)."""
    import torch
    n, _, m, d = value.shape
    lq, nl, npts = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    out = value.new_zeros((n, lq, m, d))
    start = 0
    for lvl in range(nl):
        h, w = int(value_spatial_shapes[lvl][0]), int(value_spatial_shapes[lvl][1])
        level = value[:, start:start + h * w]                                  # [N, H*W, M, D]
        start += h * w
        xy = sampling_locations[:, :, :, lvl]                                  # [N, Lq, M, P, 2]
        px, py = xy[..., 0] * w - 0.5, xy[..., 1] * h - 0.5
        x0, y0 = torch.floor(px), torch.floor(py)
        fx, fy = px - x0, py - y0
        x0, y0 = x0.long(), y0.long()
        acc = 0
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                xi, yi = x0 + dx, y0 + dy
                inside = ((xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)).to(value.dtype)
                flat = yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)             # [N, Lq, M, P]
                idx = flat.permute(0, 1, 3, 2).reshape(n, lq * npts, m, 1).expand(-1, -1, -1, d)
                rows = level.gather(1, idx).view(n, lq, npts, m, d).permute(0, 1, 3, 2, 4)
                acc = acc + (wy * wx * inside).unsqueeze(-1) * rows            # [N, Lq, M, P, D]
        out = out + (attention_weights[:, :, :, lvl].unsqueeze(-1) * acc).sum(3)
    return out.reshape(n, lq, m * d)
