"""The opt-in split-operand form of the loss nets' implicit-GEMM convolutions (csrc/conv_gemm.h: hi + lo bf16 parts, three bf16 MFMAs
per product, fp32 accumulation) against the exact fp32 form on the same inputs, through the C-ABI.  Tolerances: one layer's output
carries ~2^-17 relative per operand; 17 layers deep the embeddings stay within 3e-5 (unit vectors), the LPIPS distance within 5e-5
relative.  Gradients are compared with the ReLU decisions of ONE forward (the exact one) so that what is measured is the gradient
kernels' arithmetic, not which side of zero a rounding-sized activation fell."""
import numpy as np
import pytest
import torch

import speech2lip_amd as s2l
from speech2lip_amd import weights as W

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def syncnet(dev):
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: T(v) for k, v in W.make_syncnet_state_dict(0).items()})
    return net


def _rel(a, b):
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("B", [1, 3, 16])
def test_syncnet_split_form_embeddings_loss_and_gradient(syncnet, dev, B):
    from speech2lip_amd.syncnet import sync_window
    mel, pos, neg = (T(x).to(dev) for x in W.synthetic_sync_batch(B, seed=70 + B))
    face = torch.cat([sync_window(pos), sync_window(neg)])
    a0, v0 = (t.clone() for t in syncnet.embed_pair_nhwc(mel, face, precision="fp32"))
    a1, v1 = (t.clone() for t in syncnet.embed_pair_nhwc(mel, face, precision="split"))
    assert float((a1 - a0).abs().max()) <= 3e-5 and float((v1 - v0).abs().max()) <= 3e-5
    assert not torch.equal(v1, v0)      # (the other kernel did run)
    l0 = s2l.SyncLoss(syncnet, precision="fp32").get_sync_contrastive_loss(mel, pos, neg)
    l1 = s2l.SyncLoss(syncnet, precision="split").get_sync_contrastive_loss(mel, pos, neg)
    assert abs(float(l0) - float(l1)) <= 1e-5
    # the face encoder's input-gradient kernels in both forms over the SAME saved activations (the exact forward's)
    d = torch.nn.functional.normalize(T(np.random.default_rng(B).standard_normal((B, 512)).astype(np.float32)), dim=1).to(dev)
    syncnet.embed_pair_nhwc(mel, face, precision="fp32")
    g0 = syncnet.face_backward(d).clone()
    syncnet._last = syncnet._last[:2] + (True,)
    g1 = syncnet.face_backward(d)
    assert _rel(g1, g0) <= 5e-5, _rel(g1, g0)
    assert not torch.equal(g1, g0)


def test_syncnet_default_is_the_exact_form(syncnet, dev):
    mel, pos, neg = (T(x).to(dev) for x in W.synthetic_sync_batch(2, seed=3))
    from speech2lip_amd.syncnet import sync_window
    face = torch.cat([sync_window(pos), sync_window(neg)])
    assert syncnet.conv_precision == "fp32"
    a0, v0 = (t.clone() for t in syncnet.embed_pair_nhwc(mel, face))
    a1, v1 = syncnet.embed_pair_nhwc(mel, face, precision="fp32")
    assert torch.equal(a0, a1) and torch.equal(v0, v1)
    with pytest.raises(ValueError):
        syncnet.embed_pair_nhwc(mel, face, precision="bf16")


@pytest.mark.parametrize("shape", [(2, 96, 96), (3, 64, 80), (1, 200, 160)])
def test_lpips_split_form_distance_and_gradient(dev, shape):
    N, H, Wd = shape
    torch.manual_seed(1234)      # (the module's random weights: the linear heads may be negative, so a distance can be a small difference)
    lp = s2l.LPIPS(pretrained=False).to(dev)
    g = torch.Generator().manual_seed(N * 1000 + H)
    a, b = torch.rand(N, H, Wd, 3, generator=g).to(dev), torch.rand(N, H, Wd, 3, generator=g).to(dev)
    d0, st0 = lp.distance_nhwc(a, b, from01=True, keep=True)
    d1, st1 = lp.distance_nhwc(a, b, from01=True, keep=True, precision="split")
    assert float(((d1 - d0).abs() - 5e-5 * d0.abs()).max()) <= 3e-8      # (a distance of unit-normalised features: differences cancel)
    assert st0[-1] is False and st1[-1] is True
    w = torch.rand(N, generator=g).to(dev)
    g0 = lp.backward_nhwc(st0, w)
    g1 = lp.backward_nhwc(st0[:-1] + (True,), w)      # the split gradient kernels over the exact forward's activations
    assert _rel(g1, g0) <= 5e-5, _rel(g1, g0)
    g2 = lp.backward_nhwc(st1, w)                      # ... and the whole split pass: ReLU and max-pool decisions of its own forward
    assert _rel(g2, g0) <= 2e-2, _rel(g2, g0)          #     (a pooling window's arg-max within rounding moves a whole gradient entry)
    assert not torch.equal(g1, g0)


def test_lpips_autograd_surface_takes_the_precision(dev):
    from speech2lip_amd.autograd import lpips_distance
    torch.manual_seed(1234)
    lp = s2l.LPIPS(pretrained=False).to(dev)
    g = torch.Generator().manual_seed(5)
    a = torch.rand(2, 64, 64, 3, generator=g).to(dev).requires_grad_(True)
    b = torch.rand(2, 64, 64, 3, generator=g).to(dev)
    out = {}
    for prec in (None, "fp32", "split"):
        a.grad = None
        lpips_distance(lp, a, b, from01=True, precision=prec).sum().backward()
        out[prec] = a.grad.clone()
    assert torch.equal(out[None], out["fp32"])
    assert not torch.equal(out["split"], out["fp32"]) and _rel(out["split"], out["fp32"]) <= 2e-2
