"""Seeded input generators shared by make_golden.py (which runs the reference
on them) and by the parity tests (which re-create the same inputs and compare
against the stored outputs).  Inputs are drawn with numpy's frozen legacy
``RandomState`` streams and rounded to float32, so that the float32 and float64
tests see bit-identical values; they are *not* stored in the .npz files.

Case list follows SURVEY.md section 8(c):
  testpy_*   shapes and value distributions of the reference's ops/test.py:24-39
             (N,M,Lq,L,P = 1,2,2,2,2; levels (6,4),(3,2)); D in {2,30,32,64,71}
             hits each backward variant the reference dispatches on (test.py:88).
  enc_m8d32  pixel-decoder head shape (M=8, D=32, L=3, P=4, Lq=S), low-res level
             first, "model-like" locations (pixel-centre reference points plus
             the MSDeformAttn bias pattern and 1-px noise).
  oob        locations in [-0.5, 1.5]: zero padding and the (-1, H) gate.
  centres    locations exactly on pixel centres, on 0.0 and on 1.0.
  nonsquare  three non-square levels, odd M/D/P, catches W/H swaps.
  batch4     N=4 (exercises batch indexing / im2col_step chunking).
"""
import numpy as np


def level_start_index(shapes):
    hw = shapes[:, 0] * shapes[:, 1]
    return np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def _norm_weights(rs, n, lq, m, l, p):
    aw = rs.random_sample((n, lq, m, l, p)) + 1e-5
    return aw / aw.sum(axis=(-1, -2), keepdims=True)


def model_like_locations(rs, shapes, n, m, p, noise_px=1.0):
    """Reference points = per-level pixel-centre grid as get_reference_points
    (reference msdeformattn.py:216-229, valid_ratio 1); offsets = the
    MSDeformAttn._reset_parameters bias pattern (ops/modules/ms_deform_attn.py:68-74)
    plus N(0, noise_px) pixels, divided by (W_l, H_l) (ms_deform_attn.py:107-109)."""
    l = shapes.shape[0]
    refs = []
    for h, w in shapes:
        ys, xs = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
        refs.append(np.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = np.concatenate(refs, 0)                                   # [Lq, 2] (x, y)
    lq = ref.shape[0]
    thetas = np.arange(m, dtype=np.float64) * (2.0 * np.pi / m)
    grid = np.stack([np.cos(thetas), np.sin(thetas)], -1)
    grid = grid / np.abs(grid).max(-1, keepdims=True)                # [M, 2]
    off = grid[:, None, None, :] * (np.arange(p) + 1.0)[None, None, :, None]   # [M,1,P,2]
    off = np.broadcast_to(off, (m, l, p, 2))
    off = off[None, None] + noise_px * rs.standard_normal((n, lq, m, l, p, 2))
    wh = np.stack([shapes[:, 1], shapes[:, 0]], -1).astype(np.float64)   # (W, H)
    return ref[None, :, None, None, None, :] + off / wh[None, None, None, :, None, :]


def make_case(name):
    """Returns dict(value, shapes, lsi, loc, aw, grad_out) -- float64 arrays whose
    values are float32-representable, int64 shapes."""
    if name.startswith("testpy_d"):
        d = int(name[len("testpy_d"):])
        rs = np.random.RandomState(3 + d)
        shapes = np.array([(6, 4), (3, 2)], dtype=np.int64)
        n, m, lq, l, p = 1, 2, 2, 2, 2
        s = int((shapes[:, 0] * shapes[:, 1]).sum())
        value = rs.random_sample((n, s, m, d)) * 0.01
        loc = rs.random_sample((n, lq, m, l, p, 2))
        aw = _norm_weights(rs, n, lq, m, l, p)
    elif name == "enc_m8d32":
        rs = np.random.RandomState(101)
        shapes = np.array([(2, 2), (4, 4), (8, 8)], dtype=np.int64)
        n, m, d, p = 1, 8, 32, 4
        l = 3
        s = int((shapes[:, 0] * shapes[:, 1]).sum())
        lq = s
        value = rs.standard_normal((n, s, m, d))
        loc = model_like_locations(rs, shapes, n, m, p)
        logits = rs.standard_normal((n, lq, m, l * p))
        e = np.exp(logits - logits.max(-1, keepdims=True))
        aw = (e / e.sum(-1, keepdims=True)).reshape(n, lq, m, l, p)
    elif name == "oob":
        rs = np.random.RandomState(102)
        shapes = np.array([(5, 7), (3, 3)], dtype=np.int64)
        n, m, d, lq, l, p = 2, 2, 8, 9, 2, 3
        s = int((shapes[:, 0] * shapes[:, 1]).sum())
        value = rs.standard_normal((n, s, m, d))
        loc = rs.random_sample((n, lq, m, l, p, 2)) * 2.0 - 0.5
        aw = _norm_weights(rs, n, lq, m, l, p)
    elif name == "centres":
        rs = np.random.RandomState(103)
        shapes = np.array([(4, 4), (2, 8)], dtype=np.int64)
        n, m, d, lq, l, p = 1, 2, 4, 12, 2, 4
        s = int((shapes[:, 0] * shapes[:, 1]).sum())
        value = rs.standard_normal((n, s, m, d))
        # pixel centres of a 4x4 / 8x8 lattice (exactly representable), plus 0.0 and 1.0
        choices = np.array([0.0, 1.0, 0.125, 0.375, 0.625, 0.875, 0.0625, 0.1875, 0.5, 0.25, 0.75])
        loc = choices[rs.randint(0, len(choices), size=(n, lq, m, l, p, 2))]
        aw = _norm_weights(rs, n, lq, m, l, p)
    elif name == "nonsquare":
        rs = np.random.RandomState(104)
        shapes = np.array([(6, 4), (3, 2), (5, 7)], dtype=np.int64)
        n, m, d, lq, l, p = 2, 3, 8, 7, 3, 3
        s = int((shapes[:, 0] * shapes[:, 1]).sum())
        value = rs.standard_normal((n, s, m, d))
        loc = rs.random_sample((n, lq, m, l, p, 2)) * 1.2 - 0.1
        aw = _norm_weights(rs, n, lq, m, l, p)
    elif name == "batch4":
        rs = np.random.RandomState(105)
        shapes = np.array([(3, 5), (6, 10)], dtype=np.int64)
        n, m, d, lq, l, p = 4, 4, 32, 11, 2, 4
        s = int((shapes[:, 0] * shapes[:, 1]).sum())
        value = rs.standard_normal((n, s, m, d))
        loc = rs.random_sample((n, lq, m, l, p, 2))
        aw = _norm_weights(rs, n, lq, m, l, p)
    else:
        raise KeyError(name)
    n, s, m, d = value.shape
    lq = loc.shape[1]
    grad_out = rs.standard_normal((n, lq, m * d))
    return dict(value=_f32(value), shapes=shapes, lsi=level_start_index(shapes),
                loc=_f32(loc), aw=_f32(aw), grad_out=_f32(grad_out))


CASES = ["testpy_d2", "testpy_d30", "testpy_d32", "testpy_d64", "testpy_d71",
         "enc_m8d32", "oob", "centres", "nonsquare", "batch4"]


def make_module_case():
    """Inputs + parameters for one MSDeformAttn module call:
    d_model=32, 2 levels, 4 heads, 2 points."""
    rs = np.random.RandomState(201)
    d_model, n_levels, n_heads, n_points = 32, 2, 4, 2
    shapes = np.array([(3, 4), (6, 8)], dtype=np.int64)
    s = int((shapes[:, 0] * shapes[:, 1]).sum())
    n, lq = 2, s
    sd = {
        "sampling_offsets.weight": rs.standard_normal((n_heads * n_levels * n_points * 2, d_model)) * 0.05,
        "sampling_offsets.bias": rs.standard_normal((n_heads * n_levels * n_points * 2,)),
        "attention_weights.weight": rs.standard_normal((n_heads * n_levels * n_points, d_model)) * 0.2,
        "attention_weights.bias": rs.standard_normal((n_heads * n_levels * n_points,)) * 0.1,
        "value_proj.weight": rs.standard_normal((d_model, d_model)) * 0.2,
        "value_proj.bias": rs.standard_normal((d_model,)) * 0.1,
        "output_proj.weight": rs.standard_normal((d_model, d_model)) * 0.2,
        "output_proj.bias": rs.standard_normal((d_model,)) * 0.1,
    }
    sd = {k: _f32(v) for k, v in sd.items()}
    query = _f32(rs.standard_normal((n, lq, d_model)))
    src = _f32(rs.standard_normal((n, s, d_model)))
    ref = _f32(rs.random_sample((n, lq, n_levels, 2)))
    mask = rs.random_sample((n, s)) < 0.1
    return dict(d_model=d_model, n_levels=n_levels, n_heads=n_heads, n_points=n_points,
                shapes=shapes, lsi=level_start_index(shapes), state_dict=sd, query=query,
                src=src, ref=ref, padding_mask=mask)


# PositionEmbeddingSine fixtures: (key, constructor kwargs, input
# shape [N, C, H, W], with a padding mask?)
POSENC_CASES = [
    ("enc_norm", dict(num_pos_feats=128, normalize=True), (2, 4, 6, 9), False),
    ("enc_norm_masked", dict(num_pos_feats=128, normalize=True), (2, 4, 6, 9), True),
    ("plain", dict(num_pos_feats=16, temperature=100), (1, 3, 5, 4), False),
    ("scaled", dict(num_pos_feats=32, normalize=True, scale=3.0), (1, 2, 7, 3), True),
]


def posenc_mask(shape):
    """Padding mask [N, H, W] (True = padded): image n is valid on its top-left (H-n-1) x (W-2n-1)."""
    import torch
    n, _, h, w = shape
    mask = torch.ones((n, h, w), dtype=torch.bool)
    for i in range(n):
        mask[i, :h - i - 1, :w - 2 * i - 1] = False
    return mask


# Encoder-stack fixtures: MSDeformAttnTransformerEncoderOnly of the reference
# (pixel_decoder/msdeformattn.py:23-254).  Parameters are drawn here, by name and shape, so that the generator
# (which loads them into the reference's class) and the tests (which load them into the mirror) agree.
ENCODER_CASES = {
    "small": dict(kwargs=dict(d_model=64, nhead=4, num_encoder_layers=2, dim_feedforward=96, dropout=0.0,
                              num_feature_levels=3, enc_n_points=2, use_adapters=True, adapter_dim=16,
                              adapter_scale=0.1),
                  shapes=[(3, 4), (5, 3), (6, 7)], batch=2, seed=501),
    "heads8": dict(kwargs=dict(d_model=256, nhead=8, num_encoder_layers=2, dim_feedforward=128, dropout=0.0,
                               num_feature_levels=3, enc_n_points=4, use_adapters=False),
                   shapes=[(2, 3), (4, 5), (8, 8)], batch=1, seed=502),          # 32 channels per head: the HIP fast path
    # the reference's depth (6 layers, adapters on) at the encoder shape of a 128^2 image (strides 32/16/8)
    "six_layers": dict(kwargs=dict(d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=128, dropout=0.0,
                                   num_feature_levels=3, enc_n_points=4, use_adapters=True, adapter_dim=16,
                                   adapter_scale=0.1),
                       shapes=[(4, 4), (8, 8), (16, 16)], batch=1, seed=503),
}


def make_encoder_case(key):
    spec = ENCODER_CASES[key]
    kw, rs = spec["kwargs"], np.random.RandomState(spec["seed"])
    d, m, nl, npts = kw["d_model"], kw["nhead"], kw["num_feature_levels"], kw["enc_n_points"]
    ffn = kw["dim_feedforward"]

    def w(*shape, scale=0.1):
        return _f32(scale * rs.standard_normal(shape))

    sd = {"level_embed": w(nl, d, scale=0.5)}
    for i in range(kw["num_encoder_layers"]):
        pre = "encoder.layers.%d." % i
        sd[pre + "self_attn.sampling_offsets.weight"] = w(m * nl * npts * 2, d, scale=0.05)
        sd[pre + "self_attn.sampling_offsets.bias"] = w(m * nl * npts * 2, scale=1.5)        # pixels
        sd[pre + "self_attn.attention_weights.weight"] = w(m * nl * npts, d, scale=0.1)
        sd[pre + "self_attn.attention_weights.bias"] = w(m * nl * npts, scale=0.5)
        for name in ("value_proj", "output_proj"):
            sd[pre + "self_attn.%s.weight" % name] = w(d, d)
            sd[pre + "self_attn.%s.bias" % name] = w(d)
        for name in ("norm1", "norm2"):
            sd[pre + name + ".weight"] = _f32(1.0 + 0.1 * rs.standard_normal(d))
            sd[pre + name + ".bias"] = w(d)
        sd[pre + "linear1.weight"], sd[pre + "linear1.bias"] = w(ffn, d), w(ffn)
        sd[pre + "linear2.weight"], sd[pre + "linear2.bias"] = w(d, ffn), w(d)
        if kw.get("use_adapters"):
            ad = kw["adapter_dim"]
            sd[pre + "adapter.0.weight"], sd[pre + "adapter.0.bias"] = w(ad, d), w(ad)
            sd[pre + "adapter.2.weight"], sd[pre + "adapter.2.bias"] = w(d, ad), w(d)
    n = spec["batch"]
    srcs = [_f32(rs.standard_normal((n, d, h, ww))) for h, ww in spec["shapes"]]
    pos = [_f32(0.5 * rs.standard_normal((n, d, h, ww))) for h, ww in spec["shapes"]]
    s = sum(h * ww for h, ww in spec["shapes"])
    return dict(kwargs=kw, state_dict=sd, srcs=srcs, pos=pos, grad_memory=_f32(rs.standard_normal((n, s, d))))
