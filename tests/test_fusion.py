"""2D->3D feature fusion against golden vectors produced by the reference's OWN fusion/feat_fusion.py
(tests/golden/g7_feature_fusion.npz) -- this row's parity is PINNED to the reference, bit for bit."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_l2


def test_camera_points_and_pixels_match_reference_on_cpu(golden):
    """The part kept in PyTorch (world->camera matmul) + the kernel's pixel formula restated in numpy fp32."""
    from unipre3d_amd.fusion import FeatureFusion
    g = golden("g7_feature_fusion.npz")
    for tag in ("sq", "rect"):
        cp = FeatureFusion.camera_points(torch.tensor(g[f"{tag}_center"]), torch.tensor(g[f"{tag}_c2w"])).numpy()
        intr = g[f"{tag}_intr"]
        fx, fy, cx, cy = (np.float32(v) for v in (intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2]))
        u = np.rint((cp[..., 0] * fx) / cp[..., 2] + cx)
        v = np.rint((cp[..., 1] * fy) / cp[..., 2] + cy)
        assert np.array_equal(u.astype(np.int64), g[f"{tag}_pix"][..., 0]) and np.array_equal(v.astype(np.int64), g[f"{tag}_pix"][..., 1])
        assert np.array_equal(cp[..., 2], g[f"{tag}_depth"])


def test_fusion_oracle_is_pinned_to_the_reference_module(golden):
    """oracle/fusion_oracle.py (the checker and the bench's CPU baseline for row N4a) == the reference module's own outputs."""
    from oracle import fusion_oracle as fo
    g = golden("g7_feature_fusion.npz")
    for tag in ("sq", "rect"):
        intr = g[f"{tag}_intr"]
        cam = fo.camera_points(g[f"{tag}_center"], g[f"{tag}_c2w"])
        pix, depth = fo.pixels(cam, intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2])
        assert np.array_equal(pix, g[f"{tag}_pix"]) and np.array_equal(depth, g[f"{tag}_depth"])
        assert np.array_equal(fo.fuse(g[f"{tag}_x"], g[f"{tag}_center"], g[f"{tag}_feat"], g[f"{tag}_c2w"], intr), g[f"{tag}_out"])
        B, C, H, W = g[f"{tag}_feat"].shape
        mapped, sel = fo.mapped_features(cam, g[f"{tag}_feat"], intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2])
        N, Cx = mapped.shape[1], g[f"{tag}_x"].shape[2]
        gfeat = fo.mapped_grad(g[f"{tag}_w"][:, -N:, Cx:], sel, B, C, H, W)
        assert rel_l2(gfeat, g[f"{tag}_gfeat"]) < 1e-6


def test_fusion_library_exports():
    from unipre3d_amd import fusion
    hdr = open(os.path.join(ROOT, "include", "unipre3d_fusion.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(u3d_[a-z_0-9]+)\s*\(", hdr)))
    assert set(names) == set(fusion.EXPORTS)
    lib = fusion.load()
    null = ctypes.c_void_p(0)
    f = ctypes.c_float(1.0)
    assert lib.u3d_zbuffer_fusion_forward(1, 4, 2, 8, 8, f, f, f, f, null, null, null, null, null, null) == 1
    assert lib.u3d_zbuffer_fusion_forward(0, 4, 2, 8, 8, f, f, f, f, null, null, null, null, null, null) == 0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fusion.FeatureFusion(torch.nn.Identity())(torch.zeros(1, 4, 2), torch.zeros(1, 4, 3) + 1, torch.zeros(1, 2, 8, 8),
                                                  torch.eye(4)[None], np.eye(3, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["sq", "rect"])
def test_feature_fusion_equals_reference_output_and_gradient(golden, tag):
    from unipre3d_amd.fusion import FeatureFusion
    g = golden("g7_feature_fusion.npz")
    dev = torch.device("cuda:0")
    feat = torch.tensor(g[f"{tag}_feat"]).to(dev).requires_grad_(True)
    ff = FeatureFusion(torch.nn.Identity())
    y = ff(torch.tensor(g[f"{tag}_x"]).to(dev), torch.tensor(g[f"{tag}_center"]).to(dev), feat, torch.tensor(g[f"{tag}_c2w"]).to(dev),
           g[f"{tag}_intr"])
    assert np.array_equal(y.detach().cpu().numpy(), g[f"{tag}_out"])                 # gathered copies: bit-exact
    (y * torch.tensor(g[f"{tag}_w"]).to(dev)).sum().backward()
    assert rel_l2(feat.grad.cpu().numpy(), g[f"{tag}_gfeat"]) < 1e-6                  # scatter-add order only


@pytest.mark.gpu
def test_feature_fusion_transformer_sized():
    """Transformer config: 128 group centres, 128x128 feature map, 32 channels, batch 32; occlusion semantics checked
    against a direct numpy z-buffer."""
    from unipre3d_amd.fusion import FeatureFusion
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    B, N, C, H = 32, 128, 32, 128
    center = torch.randn(B, N, 3, generator=gen) * 0.3
    c2w = torch.eye(4).repeat(B, 1, 1); c2w[:, 3, 2] = -1.75          # row-vector form: translation in the last row
    intr = np.zeros((3, 4)); intr[0, 0] = intr[1, 1] = 140.0; intr[0, 2] = intr[1, 2] = 64.0
    feat = torch.randn(B, C, H, H, generator=gen)
    ff = FeatureFusion(torch.nn.Identity())
    mapped = ff.mapped_features(center.to(dev), feat.to(dev), c2w.to(dev), intr).cpu().numpy()
    cp = FeatureFusion.camera_points(center, c2w).numpy()
    u = np.rint((cp[..., 0] * np.float32(140)) / cp[..., 2] + np.float32(64)).astype(np.int64)
    v = np.rint((cp[..., 1] * np.float32(140)) / cp[..., 2] + np.float32(64)).astype(np.int64)
    exp = np.zeros((B, N, C), np.float32)
    for b in range(B):
        ok = (u[b] >= 0) & (v[b] >= 0) & (u[b] < H) & (v[b] < H) & (cp[b, :, 2] >= 0)
        zb = {}
        for n in np.nonzero(ok)[0]:
            k = (u[b, n], v[b, n]); zb[k] = min(zb.get(k, np.inf), cp[b, n, 2])
        for n in np.nonzero(ok)[0]:
            if cp[b, n, 2] == zb[(u[b, n], v[b, n])]:
                exp[b, n] = feat[b, :, u[b, n], v[b, n]].numpy()
    assert np.array_equal(mapped, exp) and (np.abs(exp).sum(-1) > 0).mean() > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,C", [(16, 16, 5), (9, 7, 11), (128, 128, 40)])
def test_gather_form_backward_writes_every_element_once_and_adds_tied_points(H, W, C):
    """ABI 2 backward (gather form): the kernel writes EVERY element of the (B,C,H,W) gradient (autograd hands it uninitialised memory), points that
    tie for a pixel at the same depth all contribute (fusion/feat_fusion.py:117-131), H*W need not be a multiple of 4, and the result equals the
    numpy restatement pinned to G7 (oracle/fusion_oracle.py::mapped_grad)."""
    from oracle import fusion_oracle as fo
    from unipre3d_amd import _lib, fusion
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(H * 131 + W)
    B, N = 3, 60
    cp = torch.cat([torch.randn(B, N, 2, generator=gen) * 0.4, 1.0 + torch.rand(B, N, 1, generator=gen), torch.ones(B, N, 1)], dim=-1)
    cp[:, 3] = torch.tensor([0.01, -0.02, 0.9, 1.0])      # image centre, nearer than every random point (z in [1, 2]): a certain winner
    cp[:, 10] = cp[:, 3]; cp[:, 40] = cp[:, 3]            # three points on one pixel at the SAME depth: all three are winners
    cp[:, 20, :2] = cp[:, 5, :2] * (cp[:, 20, 2:3] / cp[:, 5, 2:3])   # same pixel, farther: occluded
    f = float(min(H, W))
    feat = torch.randn(B, C, H, W, generator=gen).to(dev).requires_grad_(True)
    mapped, sel = fusion._ZBufferGather.apply(cp.to(dev), feat, f, f, H / 2.0, W / 2.0)
    w = torch.randn(B, N, C, generator=gen)
    # poison the caching allocator's free blocks so that an element the kernel skipped would show
    junk = torch.full((B, C, H, W), float("nan"), device=dev); del junk
    (mapped * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    s = sel.cpu().numpy()
    assert (s[:, 3] >= 0).all() and (s[:, 3] == s[:, 10]).all() and (s[:, 3] == s[:, 40]).all()
    ref = fo.mapped_grad(w.numpy(), s, B, C, H, W)
    got = feat.grad.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    untied = np.ones_like(ref, dtype=bool)
    for b in range(B):
        untied[b, :, s[b, 3] // W, s[b, 3] % W] = False
    assert np.array_equal(got[untied], ref[untied])         # single winners: copies, bit-exact
    # N = 0: an all-zero gradient
    gz = torch.full((1, 2, 4, 4), float("nan"), device=dev)
    assert fusion.load().u3d_zbuffer_fusion_backward(1, 0, 2, 4, 4, _lib.ptr(None), _lib.ptr(None), _lib.ptr(None), _lib.ptr(gz),
                                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    assert not gz.any()
