"""PyTorch-ROCm stand-in of the reference's object-level predictor around the hot path -- BENCH HARNESS ONLY.

The reference's backbones are out of scope ("backbone stays in PyTorch-ROCm"); this module exists so that
bench.py can time ONE end-to-end pre-training step in which every native operator of this repository sits in its real
position: FPS + ball query + grouping (N1) -> mini-PointNet tokenizer -> 16 transformer blocks -> 2D->3D feature fusion
(N4) -> `final` MLP -> fused activations + render + loss + backward (N2/N3, R4/R5) -> DDP all-reduce (R9).
It restates, with standard PyTorch layers and random initialisation, the structure (and therefore the parameter count:
encoder 29,066,880 + final 52,247 + fusion 295,296 + image 1x1 conv/GN 49,792 = 29.46 M) of
  openpoints/models/backbone/transformer.py:209-327 (PointTransformerEncoder: group_divider, Encoder, reduce_dim, cls token,
  pos_embed, 16 x Block(dim 384, 6 heads, mlp x4, qkv_bias False), LayerNorm; FeatureFusion after the last block :186-206),
  model/point_predictor.py:62-64,78-85 (num_groups 128, encoder_dims 384, depth 16; final 384-128-23),
  model/gaussian_predictor.py:205-227 (image GroupNorm + 1x1 conv 128->384, fusion Linear(768,384)+ReLU).
Not restated: the frozen SD-VAE image network (its decoder features are replaced by a synthetic (B,128,H,W) tensor) and
stochastic depth (drop_path 0).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointops
from .fusion import FeatureFusion


class _Block(nn.Module):
    def __init__(self, dim=384, heads=6, mlp_ratio=4.0):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.qkv, self.proj = nn.Linear(dim, dim * 3, bias=False), nn.Linear(dim, dim)
        self.fc1, self.fc2 = nn.Linear(dim, int(dim * mlp_ratio)), nn.Linear(int(dim * mlp_ratio), dim)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(self.norm1(x)).reshape(B, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        x = x + self.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C))
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class _Tokenizer(nn.Module):
    """transformer.py:214-242 `Encoder`: per-group mini-PointNet."""

    def __init__(self, out=384):
        super().__init__()
        self.first = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True), nn.Conv1d(128, 256, 1))
        self.second = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, out, 1))

    def forward(self, groups):                      # (B, G, K, 3)
        B, G, K, _ = groups.shape
        f = self.first(groups.reshape(B * G, K, 3).transpose(2, 1))
        f = torch.cat([f.max(dim=2, keepdim=True)[0].expand(-1, -1, K), f], dim=1)
        return self.second(f).max(dim=2)[0].reshape(B, G, -1)


class PointTransformerStandIn(nn.Module):
    def __init__(self, num_groups=128, group_size=32, radius=0.1, dim=384, depth=16, heads=6, image_in=128, out_channels=23):
        super().__init__()
        self.num_groups = num_groups
        self.grouper = pointops.QueryAndGroup(radius, group_size)
        self.encoder = _Tokenizer(dim)
        self.reduce_dim = nn.Linear(dim, dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, dim))
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, dim))
        self.blocks = nn.ModuleList([_Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)
        self.image_conv = nn.Sequential(nn.GroupNorm(32, image_in, eps=1e-6), nn.Conv2d(image_in, dim, kernel_size=1))
        self.fusion_mlps = nn.Sequential(nn.Linear(dim + dim, dim), nn.ReLU())
        self.final = nn.Sequential(nn.Linear(dim, 128), nn.ReLU(), nn.Linear(128, out_channels))

    def encoder_parameters(self):
        mods = (self.encoder, self.reduce_dim, self.pos_embed, self.blocks, self.norm)
        return sum(p.numel() for m in mods for p in m.parameters()) + self.cls_token.numel() + self.cls_pos.numel()

    def forward(self, pts, image_decoder_features, c2w, intrinsic):
        """pts (B,N,3), image_decoder_features (B,128,H,W), c2w (B,4,4) row-vector view->world, intrinsic (3,4) numpy.
        Returns (head_out (B,G,23) point-major, centers (B,G,3))."""
        pts = pts.contiguous()
        idx = pointops.furthest_point_sample(pts, self.num_groups).long()
        center = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        grouped, _ = self.grouper(center, pts)                                   # (B,3,G,K) relative coordinates
        tok = self.reduce_dim(self.encoder(grouped.permute(0, 2, 3, 1)))
        x = torch.cat([self.cls_token.expand(tok.shape[0], -1, -1), tok], dim=1)
        pos = torch.cat([self.cls_pos.expand(tok.shape[0], -1, -1), self.pos_embed(center)], dim=1)
        fuse = FeatureFusion(self.fusion_mlps)
        img = self.image_conv(image_decoder_features)
        for i, blk in enumerate(self.blocks):
            x = blk(x + pos)
            if i == len(self.blocks) - 1:
                x = fuse(x, center, img, c2w, intrinsic)
        x = self.norm(x)[:, 1:, :]
        return self.final(x), center


def object_intrinsics(fov_deg: float, res: int) -> np.ndarray:
    """model/gaussian_predictor.py:263-277."""
    k = np.zeros((3, 4))
    k[2, 2] = 1
    k[0, 0] = k[1, 1] = (res / 2.0) / np.tan(np.radians(fov_deg / 2.0))
    k[0, 2] = k[1, 2] = res / 2.0
    return k
