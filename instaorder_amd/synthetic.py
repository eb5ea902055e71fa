"""Synthetic MSDA inputs of the pixel-decoder encoder's shape.

Shapes follow the reference's encoder: levels res5/res4/res3 (strides 32/16/8), LOW-RES
FIRST (reference msdeformattn.py:440), encoder self-attention so Lq == S, M=8 heads of D=32,
P=4 points (msdeformattn.py:33; configs/instaorder/maskformer2_R50_bs16_50ep.yaml:10-29).

Two location distributions:
  "model"   reference points = per-level pixel-centre grid exactly as get_reference_points
            (msdeformattn.py:216-229, valid ratio 1) broadcast to every level, plus the
            MSDeformAttn._reset_parameters offset pattern (8 unit-max-norm directions x (p+1)
            pixels, ops/modules/ms_deform_attn.py:68-74) plus N(0, 1 px) noise, divided by
            (W_l, H_l) (ms_deform_attn.py:107-109);
  "uniform" U[0,1) as ops/test.py:37 -- worst-case locality.
Everything is drawn on the CPU generator (seeded) and then moved to `device`.
"""
import math

import torch


def encoder_shapes(image_size, strides=(32, 16, 8)):
    """[(H, W)] per level, low-res first, for a square (int) or (H, W) input."""
    if isinstance(image_size, int):
        image_size = (image_size, image_size)
    return [(math.ceil(image_size[0] / s), math.ceil(image_size[1] / s)) for s in strides]


def level_start_index(spatial_shapes):
    hw = spatial_shapes[:, 0] * spatial_shapes[:, 1]
    return torch.cat((hw.new_zeros((1,)), hw.cumsum(0)[:-1]))


def reference_points(shapes):
    """[S, 2] (x, y) pixel centres of every level, normalised -- the encoder's queries."""
    pts = []
    for h, w in shapes:
        ys, xs = torch.meshgrid((torch.arange(h, dtype=torch.float32) + 0.5) / h,
                                (torch.arange(w, dtype=torch.float32) + 0.5) / w, indexing="ij")
        pts.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1))
    return torch.cat(pts, 0)


def make_inputs(batch, shapes, heads=8, channels=32, points=4, dist="model", seed=0,
                dtype=torch.float32, device="cpu", num_query=None, noise_px=1.0):
    """Returns dict(value, spatial_shapes, level_start_index, sampling_locations,
    attention_weights, grad_output) on `device`."""
    g = torch.Generator().manual_seed(seed)
    shapes = [(int(h), int(w)) for h, w in shapes]
    nl = len(shapes)
    s = sum(h * w for h, w in shapes)
    lq = s if num_query is None else int(num_query)
    value = torch.randn((batch, s, heads, channels), generator=g)
    grad_output = torch.randn((batch, lq, heads * channels), generator=g)
    attn = torch.softmax(torch.randn((batch, lq, heads, nl * points), generator=g), -1)
    attn = attn.view(batch, lq, heads, nl, points)
    if dist == "uniform":
        loc = torch.rand((batch, lq, heads, nl, points, 2), generator=g)
    elif dist == "model":
        if lq != s:
            raise ValueError("model-like locations need num_query == S (encoder self-attention)")
        ref = reference_points(shapes)                                   # [S, 2]
        thetas = torch.arange(heads, dtype=torch.float32) * (2.0 * math.pi / heads)
        dirs = torch.stack((thetas.cos(), thetas.sin()), -1)
        dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]                # [M, 2]
        off = dirs[:, None, None, :] * torch.arange(1, points + 1, dtype=torch.float32)[None, None, :, None]
        off = off.expand(heads, nl, points, 2)[None, None]
        off = off + noise_px * torch.randn((batch, lq, heads, nl, points, 2), generator=g)
        wh = torch.tensor([(w, h) for h, w in shapes], dtype=torch.float32)
        loc = ref[None, :, None, None, None, :] + off / wh[None, None, None, :, None, :]
    else:
        raise ValueError(dist)
    ss = torch.tensor(shapes, dtype=torch.int64)
    out = dict(value=value.to(dtype), spatial_shapes=ss, level_start_index=level_start_index(ss),
               sampling_locations=loc.to(dtype), attention_weights=attn.to(dtype),
               grad_output=grad_output.to(dtype))
    return {k: v.to(device).contiguous() for k, v in out.items()}


def num_samples(batch, shapes, heads=8, points=4, num_query=None):
    s = sum(int(h) * int(w) for h, w in shapes)
    lq = s if num_query is None else num_query
    return batch * lq * heads * len(shapes) * points


def algorithmic_bytes_per_sample(levels=3, points=4, channels=32, value_bytes=4, loc_bytes=4):
    """(fwd, bwd) algorithmic HBM bytes per sample when Lq == S: every input read once,
    every output written once: 33.33 / 56.00 for fp32, L*P = 12."""
    lp = levels * points
    row = channels * value_bytes
    fwd = (row + row) / lp + 3 * loc_bytes
    bwd = (row + row + row) / lp + 3 * loc_bytes + 3 * loc_bytes
    return fwd, bwd
