"""``MSDeformAttnPixelDecoder`` without detectron2: the rest of the
pixel decoder around the MSDA encoder -- 1x1 input projections + GroupNorm, the FPN top-down
path for the strides the encoder does not take, and the mask-feature head -- restated from
reference mask2former/modeling/pixel_decoder/msdeformattn.py:257-487 with the same sub-module
names (``input_proj``, ``transformer``, ``pe_layer``, ``mask_features``, ``adapter_i``,
``layer_i``), so the pixel-decoder part of a reference checkpoint loads unchanged.

Everything here is stock PyTorch (MIOpen convolutions, hipBLASLt); only the encoder's sampling op is
ours.  detectron2's ``ShapeSpec`` / ``Conv2d`` / ``get_norm`` / registry are replaced by the few
lines they amount to for this module.  Parity: the encoder inside is pinned (encoder.py); the convolution /
GroupNorm / FPN glue around it is unpinned -- the reference class EXECUTES detectron2's Conv2d and get_norm,
which are not installable here, and a stand-in for them would pin nothing -- so its tests check shapes, dtype
behaviour, state-dict keys and the GPU (HIP kernels) against the CPU path."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .encoder import MSDeformAttnTransformerEncoderOnly, PositionEmbeddingSine


class NormConv2d(nn.Conv2d):
    """nn.Conv2d followed by an optional norm layer and activation, stored as attributes of the
    conv (detectron2's Conv2d wrapper; keys ``weight``, ``bias``, ``norm.weight``, ``norm.bias``)."""

    def __init__(self, *args, norm=None, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = super().forward(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def _get_norm(norm, channels):
    """"" -> none, "GN" -> GroupNorm(32) (the two settings the reference's configs use)."""
    if norm in (None, ""):
        return None
    if norm == "GN":
        return nn.GroupNorm(32, channels)
    if callable(norm):
        return norm(channels)
    raise ValueError("unsupported norm %r (use '' or 'GN' or a callable)" % (norm,))


def _c2_xavier_fill(conv):
    nn.init.kaiming_uniform_(conv.weight, a=1)
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


class MSDeformAttnPixelDecoder(nn.Module):
    """input_shape: {name: (channels, stride)} of the backbone features, e.g. Swin-T
    {"res2": (96, 4), "res3": (192, 8), "res4": (384, 16), "res5": (768, 32)}."""

    def __init__(self, input_shape, *, transformer_dropout=0.0, transformer_nheads=8,
                 transformer_dim_feedforward=1024, transformer_enc_layers=6, conv_dim=256, mask_dim=256,
                 norm="GN", transformer_in_features=("res3", "res4", "res5"), common_stride=4,
                 use_adapters=True, adapter_dim=64, adapter_scale=0.1):
        super().__init__()
        shapes = sorted(input_shape.items(), key=lambda kv: kv[1][1])
        self.in_features = [k for k, _ in shapes]                       # res2 .. res5
        self.feature_strides = [v[1] for _, v in shapes]
        self.feature_channels = [v[0] for _, v in shapes]
        tshapes = [(k, v) for k, v in shapes if k in transformer_in_features]
        self.transformer_in_features = [k for k, _ in tshapes]
        transformer_in_channels = [v[0] for _, v in tshapes]
        self.transformer_feature_strides = [v[1] for _, v in tshapes]
        self.transformer_num_feature_levels = len(self.transformer_in_features)
        chans = transformer_in_channels[::-1] if self.transformer_num_feature_levels > 1 \
            else [transformer_in_channels[-1]]
        self.input_proj = nn.ModuleList(
            nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim)) for c in chans)
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels, use_adapters=use_adapters,
            adapter_dim=adapter_dim, adapter_scale=adapter_scale)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)
        self.mask_dim = mask_dim
        self.mask_features = NormConv2d(conv_dim, mask_dim, kernel_size=1, stride=1, padding=0)
        _c2_xavier_fill(self.mask_features)
        self.maskformer_num_feature_levels = 3                          # always 3 scales
        self.common_stride = common_stride
        stride = min(self.transformer_feature_strides)
        self.num_fpn_levels = int(math.log2(stride) - math.log2(self.common_stride))
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(self.feature_channels[:self.num_fpn_levels]):
            lateral = NormConv2d(in_channels, conv_dim, kernel_size=1, bias=use_bias,
                                 norm=_get_norm(norm, conv_dim))
            output = NormConv2d(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                norm=_get_norm(norm, conv_dim), activation=F.relu)
            _c2_xavier_fill(lateral)
            _c2_xavier_fill(output)
            self.add_module("adapter_{}".format(idx + 1), lateral)
            self.add_module("layer_{}".format(idx + 1), output)
            lateral_convs.append(lateral)
            output_convs.append(output)
        self.lateral_convs = lateral_convs[::-1]                        # top-down order
        self.output_convs = output_convs[::-1]

    def forward_features(self, features):
        """features: {name: [N, C, H, W]} -> (mask_features, lowest-resolution encoder output,
        [3 multi-scale features, low resolution first]).  Runs in fp32 like the reference
        (msdeformattn.py:435-441: autocast off, inputs cast to float)."""
        with torch.autocast(device_type=next(iter(features.values())).device.type, enabled=False):
            srcs, pos = [], []
            for idx, f in enumerate(self.transformer_in_features[::-1]):
                x = features[f].float()
                srcs.append(self.input_proj[idx](x))
                pos.append(self.pe_layer(x))
            y, spatial_shapes, level_start_index = self.transformer(srcs, pos)
            bs = y.shape[0]
            sizes = [int(h) * int(w) for h, w in spatial_shapes.tolist()]
            out = [z.transpose(1, 2).reshape(bs, -1, int(h), int(w))
                   for z, (h, w) in zip(torch.split(y, sizes, dim=1), spatial_shapes.tolist())]
            for idx, f in enumerate(self.in_features[:self.num_fpn_levels][::-1]):
                cur_fpn = self.lateral_convs[idx](features[f].float())
                y = cur_fpn + F.interpolate(out[-1], size=cur_fpn.shape[-2:], mode="bilinear",
                                            align_corners=False)
                out.append(self.output_convs[idx](y))
            multi_scale_features = out[:self.maskformer_num_feature_levels]
            return self.mask_features(out[-1]), out[0], multi_scale_features
