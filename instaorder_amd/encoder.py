"""The MSDA encoder stack of Mask2Former's pixel decoder, without detectron2 -- the immediate
caller of the op.  Mirrors, with the same sub-module names (so
state-dict keys match), the reference's

  MSDeformAttnTransformerEncoderOnly / ...EncoderLayer / ...Encoder
      mask2former/modeling/pixel_decoder/msdeformattn.py:23-254
  PositionEmbeddingSine
      mask2former/modeling/transformer_decoder/position_encoding.py:12-56

Levels are passed LOW-RES FIRST, as MSDeformAttnPixelDecoder does (msdeformattn.py:440).
Parity: pinned.  MSDeformAttn by tests/golden/module_msdeformattn.npz; the stack by
tests/golden/encoder_{small,heads8,six_layers}.npz and position_encoding.npz -- outputs and input gradients of the
REFERENCE's own classes, which tests/golden/make_golden.py imports with name-only placeholders for the detectron2
symbols their module merely imports (nothing the encoder classes execute comes from a placeholder)."""
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import normal_

from .modules import MSDeformAttn


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, x, mask=None):
        if mask is None:
            mask = torch.zeros((x.size(0), x.size(2), x.size(3)), device=x.device, dtype=torch.bool)
        keep = ~mask
        y_embed = keep.cumsum(1, dtype=torch.float32)
        x_embed = keep.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_pos_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def _activation(name):
    return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[name]


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4,
                 n_heads=8, n_points=4, use_adapters=False, adapter_dim=64, adapter_scale=0.1):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.use_adapters, self.adapter_scale = use_adapters, adapter_scale
        if use_adapters:        # InstaFormer's trainable adapters (msdeformattn.py:155-166)
            self.adapter = nn.Sequential(nn.Linear(d_model, adapter_dim), nn.ReLU(),
                                         nn.Linear(adapter_dim, d_model))
            nn.init.kaiming_normal_(self.adapter[0].weight)
            nn.init.zeros_(self.adapter[0].bias)
            nn.init.zeros_(self.adapter[2].weight)
            nn.init.zeros_(self.adapter[2].bias)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        query = src if pos is None else src + pos
        src = self.norm1(src + self.dropout1(self.self_attn(
            query, reference_points, src, spatial_shapes, level_start_index, padding_mask)))
        side = self.adapter(src) if self.use_adapters else None
        ffn = self.linear2(self.dropout2(self.activation(self.linear1(src))))
        src = self.norm2(src + self.dropout3(ffn))
        if side is not None:
            src = src + self.adapter_scale * side
        return src


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel centres of every level, normalised by the valid extent, then expressed in every
        level's frame: [N, S, L, 2] (msdeformattn.py:215-229)."""
        refs = []
        for lvl, (h, w) in enumerate(spatial_shapes):
            h, w = int(h), int(w)
            ys, xs = torch.meshgrid(
                torch.linspace(0.5, h - 0.5, h, dtype=torch.float32, device=device),
                torch.linspace(0.5, w - 0.5, w, dtype=torch.float32, device=device), indexing="ij")
            ys = ys.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * h)
            xs = xs.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * w)
            refs.append(torch.stack((xs, ys), -1))
        return torch.cat(refs, 1)[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None):
        reference_points = self.get_reference_points(spatial_shapes, valid_ratios, device=src.device)
        for layer in self.layers:
            src = layer(src, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return src


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", num_feature_levels=4, enc_n_points=4, use_adapters=False,
                 adapter_dim=64, adapter_scale=0.1):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        layer = MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                    num_feature_levels, nhead, enc_n_points,
                                                    use_adapters, adapter_dim, adapter_scale)
        self.encoder = MSDeformAttnTransformerEncoder(layer, num_encoder_layers)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for mod in self.modules():
            if isinstance(mod, MSDeformAttn):
                mod._reset_parameters()
        normal_(self.level_embed)

    @staticmethod
    def get_valid_ratio(mask):
        _, h, w = mask.shape
        valid_h = torch.sum(~mask[:, :, 0], 1).float() / h
        valid_w = torch.sum(~mask[:, 0, :], 1).float() / w
        return torch.stack([valid_w, valid_h], -1)

    def forward(self, srcs, pos_embeds):
        """srcs / pos_embeds: per level [N, C, H_l, W_l].  Returns (memory [N,S,C],
        spatial_shapes [L,2], level_start_index [L])  (msdeformattn.py:79-123)."""
        masks = [torch.zeros((x.size(0), x.size(2), x.size(3)), device=x.device, dtype=torch.bool)
                 for x in srcs]
        flat, flat_mask, flat_pos, shapes = [], [], [], []
        for lvl, (src, mask, pos) in enumerate(zip(srcs, masks, pos_embeds)):
            shapes.append(tuple(src.shape[2:]))
            flat.append(src.flatten(2).transpose(1, 2))
            flat_mask.append(mask.flatten(1))
            flat_pos.append(pos.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1))
        flat, flat_mask, flat_pos = torch.cat(flat, 1), torch.cat(flat_mask, 1), torch.cat(flat_pos, 1)
        spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=flat.device)
        level_start_index = torch.cat((spatial_shapes.new_zeros((1,)),
                                       spatial_shapes.prod(1).cumsum(0)[:-1]))
        valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
        memory = self.encoder(flat, spatial_shapes, level_start_index, valid_ratios, flat_pos, flat_mask)
        return memory, spatial_shapes, level_start_index
