"""PyTorch (CPU) restatement of the reference's slow path -- TEST INFRASTRUCTURE ONLY.

Restates ``ms_deform_attn_core_pytorch``
(reference: mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:52-72):
one ``grid_sample`` (bilinear, zero padding, align_corners=False) per level over
the level's value map viewed as [N*M, D, H, W], sampling at ``2*loc - 1``, then
a weighted sum over the L*P samples.  Gradients come from autograd.

It is the "port" timed by bench.py's cpu_baseline leg (the reference runs
exactly this maths on a CPU-only host through the bare ``except`` in
ops/modules/ms_deform_attn.py:116-121) and a second checker beside the C
oracle.  Pinned by tests/golden/.  Never imported by the product path.
"""
import torch
import torch.nn.functional as F


def msda_forward_torch(value, spatial_shapes, sampling_locations, attention_weights):
    n, _, m, d = value.shape
    lq, nl, npts = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    hw = [(int(h), int(w)) for h, w in spatial_shapes]
    per_level = value.split([h * w for h, w in hw], dim=1)
    grid = sampling_locations * 2 - 1                       # [-1, 1] convention of grid_sample
    sampled = []
    for lvl, (h, w) in enumerate(hw):
        fmap = per_level[lvl].permute(0, 2, 3, 1).reshape(n * m, d, h, w)
        g = grid[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(n * m, lq, npts, 2)
        sampled.append(F.grid_sample(fmap, g, mode="bilinear", padding_mode="zeros",
                                     align_corners=False))   # [N*M, D, Lq, P]
    sampled = torch.stack(sampled, dim=3).reshape(n * m, d, lq, nl * npts)
    wts = attention_weights.permute(0, 2, 1, 3, 4).reshape(n * m, 1, lq, nl * npts)
    out = (sampled * wts).sum(-1)                            # [N*M, D, Lq]
    return out.reshape(n, m * d, lq).transpose(1, 2).contiguous()


def msda_forward_backward_torch(value, spatial_shapes, sampling_locations, attention_weights, grad_output):
    """Returns (out, grad_value, grad_sampling_loc, grad_attn_weight) via autograd."""
    v = value.detach().clone().requires_grad_(True)
    loc = sampling_locations.detach().clone().requires_grad_(True)
    aw = attention_weights.detach().clone().requires_grad_(True)
    out = msda_forward_torch(v, spatial_shapes, loc, aw)
    out.backward(grad_output)
    return out.detach(), v.grad, loc.grad, aw.grad
