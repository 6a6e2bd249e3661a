"""oracle/fusion_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy, fp32) of the projection -> z-buffer -> gather chain of the reference's object-level 2D->3D fusion,
`FeatureFusion.__call__` (fusion/feat_fusion.py:58-131), and of the gradient autograd derives for it.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

PINNED: tests/test_fusion.py checks it bit for bit against golden G7 (tests/golden/g7_feature_fusion.npz), which holds the outputs
and gradients of the reference's OWN module run in the build container (tests/golden/make_golden.py::g7_feature_fusion).

Line map:
  camera_points   feat_fusion.py:37-45   homogeneous coordinate, w2c = inv(c2w^T), matmul
  pixels          feat_fusion.py:47-56   (x fx) / z + cx, (y fy) / z + cy, round half to even, depth = z
  mapped_features feat_fusion.py:90-131  inside test WITH the reference's H/W swap (:91-97), ids = b HW + py H + px (:108),
                                         amin z-buffer (:110-114), `==` test keeps ALL tied points (:117), gather [b, :, px, py] (:123-131)
  mapped_grad     what autograd gives for the index_put / index chain: grad_features[b, :, px, py] += grad_mapped[b, n, :]
"""
import numpy as np


def camera_points(center, c2w):
    """(B,N,3) world points, (B,4,4) camera-to-world (row-vector form) -> (B,N,4) camera-space points, fp32.  Evaluated with the
    reference's own three tensor operations on the CPU (torch.linalg.inv, matmul): numpy's LAPACK / BLAS round differently in the last
    bit of the depth, and depth ties decide the z-test."""
    import torch
    center = torch.as_tensor(np.asarray(center, np.float32))
    hom = torch.cat([center, torch.ones([*center.shape[:2], 1])], dim=2)
    w2c = torch.linalg.inv(torch.as_tensor(np.asarray(c2w, np.float32)).permute(0, 2, 1))
    return torch.matmul(w2c, hom.transpose(1, 2)).transpose(1, 2).contiguous().numpy()


def pixels(cam, fx, fy, cx, cy):
    """-> (pix (B,N,2) int64, depth (B,N) fp32).  Every operation is rounded to fp32 like the reference's tensor arithmetic."""
    cam = np.asarray(cam, np.float32)
    fx, fy, cx, cy = (np.float32(v) for v in (fx, fy, cx, cy))
    with np.errstate(divide="ignore", invalid="ignore"):
        u = np.rint((cam[..., 0] * fx) / cam[..., 2] + cx)
        v = np.rint((cam[..., 1] * fy) / cam[..., 2] + cy)
    big = np.float32(1e9)
    u = np.where(np.abs(u) < big, u, np.float32(-1))          # NaN / inf / beyond any image: outside (torch's .long() is UB there)
    v = np.where(np.abs(v) < big, v, np.float32(-1))
    return np.stack([u, v], axis=-1).astype(np.int64), cam[..., 2]


def mapped_features(cam, feat, fx, fy, cx, cy):
    """cam (B,N,4), feat (B,C,H,W) -> (mapped (B,N,C) fp32, sel (B,N) int32 = px*W + py of the winners, -1 elsewhere)."""
    feat = np.asarray(feat, np.float32)
    B, C, H, W = feat.shape
    pix, depth = pixels(cam, fx, fy, cx, cy)
    N = pix.shape[1]
    px, py = pix[..., 0], pix[..., 1]
    inside = (px >= 0) & (py >= 0) & (px < H) & (py < W) & (depth >= 0)
    bi, ni = np.nonzero(inside)
    x, y, d = px[bi, ni], py[bi, ni], depth[bi, ni]
    ids = bi * (H * W) + y * H + x
    zmin = np.full(B * H * W + 1, np.inf, np.float32)
    np.minimum.at(zmin, ids, d)
    win = d == zmin[ids]
    mapped = np.zeros((B, N, C), np.float32)
    sel = np.full((B, N), -1, np.int32)
    bw, nw, xw, yw = bi[win], ni[win], x[win], y[win]
    mapped[bw, nw] = feat[bw, :, xw, yw]
    sel[bw, nw] = (xw * W + yw).astype(np.int32)
    return mapped, sel


def mapped_grad(grad_mapped, sel, B, C, H, W):
    """Gradient w.r.t. the feature map: scatter-add of the winners' rows (fp32 accumulation in point order)."""
    g = np.zeros((B, C, H * W), np.float32)
    gm = np.asarray(grad_mapped, np.float32)
    for b in range(B):
        n = np.nonzero(sel[b] >= 0)[0]
        np.add.at(g[b].T, sel[b, n], gm[b, n])
    return g.reshape(B, C, H, W)


def fuse(x, center, feat, c2w, intr):
    """The whole call with an identity fusion MLP (feat_fusion.py:58-145): -> (B, N[+1], Cx + C)."""
    x = np.asarray(x, np.float32)
    if np.asarray(c2w).ndim == 4:
        c2w = np.asarray(c2w)[:, 0]
    B, N = np.asarray(center).shape[:2]
    C = np.asarray(feat).shape[1]
    mapped, _ = mapped_features(camera_points(center, c2w), feat, intr[0][0], intr[1][1], intr[0][2], intr[1][2])
    if x.shape[1] > N:                                          # transformer CLS token (:134-140)
        patch = np.concatenate([x[:, 1:], mapped], axis=-1)
        cls = np.concatenate([x[:, 0:1], np.zeros((B, 1, C), np.float32)], axis=-1)
        return np.concatenate([cls, patch], axis=1)
    return np.concatenate([x, mapped], axis=-1)
