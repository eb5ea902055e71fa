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
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None
