"""GPU parity of the perceptual term: LPIPS(net='alex', version='0.1') forward and the gradient with respect
to the generated image, against oracle/s2l_oracle.py's restatement of the lpips==0.1.4 package on the same seeded weights
(structural parity: the package and its weights are not available -- see the oracle's header), through the C-ABI
(s2l_lpips_*), the module with the package's signature, the autograd wrapper and Trainer.add_perceptual_loss."""
import numpy as np
import pytest
import torch

import speech2lip_amd as s2l
from oracle import s2l_oracle as O
from speech2lip_amd import weights as W

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lp(dev):
    m = s2l.LPIPS(net="alex", version="0.1").to(dev)
    m.load_state_dict({k: T(v) for k, v in W.make_lpips_state_dict(0).items()}, strict=True)
    return m


@pytest.fixture(scope="module")
def sd():
    return O.to_sd(W.make_lpips_state_dict(0), torch.float64)


def images(n, h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    base = 0.5 + 0.3 * np.sin(7 * xx + 2 * yy)[None, :, :, None]
    a = np.clip(base + 0.5 * (rng.random((n, h, w, 3)) - 0.5), 0, 1).astype(np.float32)
    b = np.clip(base + 0.5 * (rng.random((n, h, w, 3)) - 0.5), 0, 1).astype(np.float32)
    return T(a), T(b)


def relerr(a, b):
    b = torch.as_tensor(b).detach()
    return float((a.detach().cpu().double() - b.double()).abs().max()) / (float(b.abs().max()) + 1e-30)


@pytest.mark.parametrize("n,h,w", [(1, 96, 96), (3, 64, 83), (2, 31, 31), (1, 500, 500), (2, 131, 277)])
def test_lpips_forward_and_gradient_vs_oracle(lp, sd, dev, n, h, w):
    """distance [N] and d distance / d in0 against fp64 autograd through the oracle: the lip size (96x96, training.py:420-421),
    a ragged batch, the smallest image AlexNet's poolings admit, the fused face (500x500, :453-456), and a size whose conv1 tiles
    (16 x 16 outputs; 8 x 16 blocks of 4 x 4 input pixels in the gradient) are ragged on both axes."""
    a, b = images(n, h, w, 5 + h)
    x0 = (a.double() * 2 - 1).permute(0, 3, 1, 2).requires_grad_(True)
    x1 = (b.double() * 2 - 1).permute(0, 3, 1, 2)
    ref = O.lpips_alex(sd, x0, x1).flatten()
    wgt = torch.linspace(0.5, 1.5, n, dtype=torch.float64)
    (ref * wgt).sum().backward()
    out, state = lp.distance_nhwc((a * 2 - 1).to(dev), (b * 2 - 1).to(dev), keep=True)
    assert float(ref.detach().min()) > 1e-3
    assert relerr(out, ref) <= 2e-5, relerr(out, ref)
    g = lp.backward_nhwc(state, wgt.float().to(dev))
    gref = x0.grad.permute(0, 2, 3, 1)
    assert float(gref.abs().max()) > 0
    assert relerr(g, gref) <= 2e-4, relerr(g, gref)
    # [0,1] inputs with the reference's (x - 0.5) * 2 folded in: same distance, gradient times the chain factor; accumulation
    out01, st01 = lp.distance_nhwc(a.to(dev), b.to(dev), from01=True, keep=True)
    assert relerr(out01, ref) <= 2e-5
    base = (torch.rand(n, h, w, 3, device=dev) - 0.5) * float(gref.abs().max())
    acc = base.clone()
    lp.backward_nhwc(st01, wgt.float().to(dev), out=acc)
    assert relerr(acc - base, 2 * gref) <= 4e-4


def test_lpips_module_signature_autograd_and_trainer(lp, sd, dev):
    """The package's call -- LPIPS(in0, in1) on NCHW images in [-1,1] -> [N,1,1,1], differentiable -- and
    Trainer.add_perceptual_loss (training.py:655-674) driven like the reference drives it, against the oracle."""
    a, b = images(2, 96, 96, 3)
    pred = a.to(dev).requires_grad_(True)
    d = lp((pred.permute(0, 3, 1, 2) - 0.5) * 2, (b.to(dev).permute(0, 3, 1, 2) - 0.5) * 2)
    assert tuple(d.shape) == (2, 1, 1, 1)
    d.mean().backward()
    po = a.double().requires_grad_(True)
    ref = O.perceptual_loss(sd, po, b.double(), 1.0)
    ref.backward()
    assert relerr(d.mean(), ref) <= 2e-5
    assert relerr(pred.grad, po.grad) <= 2e-4
    # identical images: distance exactly 0
    assert float(lp(pred.detach().permute(0, 3, 1, 2), pred.detach().permute(0, 3, 1, 2)).abs().max()) == 0.0
    # the Trainer method, two terms pending at once (lip and face), as in one iteration of train_stage1
    tr = s2l.Trainer.__new__(s2l.Trainer)
    tr.perceptual_loss_fn = lp
    a2, b2 = images(1, 120, 100, 9)
    p1, p2 = a.to(dev).requires_grad_(True), a2.to(dev).requires_grad_(True)
    loss = {"loss": 0, "loss_perceptual": 0}
    tr.add_perceptual_loss(p1, b.to(dev), loss, weights=0.01)
    tr.add_perceptual_loss(p2, b2.to(dev), loss, mask=torch.ones(1, 3, 120, 100, device=dev), weights=0.02)
    loss["loss"].backward()
    q1, q2 = a.double().requires_grad_(True), a2.double().requires_grad_(True)
    ro = O.perceptual_loss(sd, q1, b.double(), 0.01) + O.perceptual_loss(sd, q2, b2.double(), 0.02)
    ro.backward()
    assert relerr(loss["loss"], ro) <= 2e-5 and relerr(loss["loss_perceptual"], ro) <= 2e-5
    assert relerr(p1.grad, q1.grad) <= 2e-4 and relerr(p2.grad, q2.grad) <= 2e-4


def test_lpips_errors(lp, dev):
    with pytest.raises(ValueError, match="at least 31x31"):
        lp.distance_nhwc(torch.zeros(1, 30, 64, 3, device=dev), torch.zeros(1, 30, 64, 3, device=dev))
    with pytest.raises(s2l._abi.S2LError, match="no CPU fallback"):
        lp.distance_nhwc(torch.zeros(1, 64, 64, 3), torch.zeros(1, 64, 64, 3))
    with pytest.raises(NotImplementedError):
        s2l.LPIPS(net="vgg")
    assert lp.distance_nhwc(torch.zeros(0, 64, 64, 3, device=dev), torch.zeros(0, 64, 64, 3, device=dev)).numel() == 0


def test_stage_one_step_with_perceptual_terms_vs_oracle_autograd(lp, dev):
    """StageOneStep(perceptual=LPIPS): loss = MSE(lip) + w LPIPS(lip) + MSE(face) + w LPIPS(face) through composite + frozen
    U-Net (training.py:417-459), every MLP gradient against torch autograd through the oracle (fp32 mode)."""
    from tests.test_gpu_parity import make_model
    h, w, B, FH, FW, x0, y0 = 32, 40, 2, 64, 72, 14, 20
    np_sd, np_unet = W.make_state_dict(0, "he"), W.make_unet_state_dict(0)
    m = make_model(dev, h, w)
    m.load_state_dict({k: T(v) for k, v in np_unet.items()})
    m.post_fusion_unet.eval()
    rng = np.random.default_rng(4)
    win = T(W.synthetic_audio(B, seed=7).astype(np.float32))
    idx, u01 = [3, 40], [0.3, 0.9]
    targets = T(rng.random((B, h * w, 3), dtype=np.float32))
    face = T(rng.random((1, FH, FW, 3), dtype=np.float32))
    gt = T(rng.random((B, FH, FW, 3), dtype=np.float32))
    mask = torch.zeros(1, FH, FW, 3)
    mask[:, y0:y0 + h, x0:x0 + w] = 1
    coord = T(W.synthetic_warp_coords(B, FH, FW, seed=3))
    step = s2l.StageOneStep(m, h, w, precision="fp32", face_loss=True, perceptual=lp, w_perceptual_loss=0.05)
    loss, g, aux = step.loss_and_grads(win.to(dev), idx, targets.to(dev), u01,
                                       face=dict(rgb_face_canonical=face.to(dev), rgb_face_gt=gt.to(dev), mask_lip_canonical=mask.to(dev),
                                                 lip_lefttop_x=x0, lip_lefttop_y=y0, coord=coord.to(dev)))
    sd_ = {k: T(v).clone().requires_grad_(True) for k, v in np_sd.items()}
    usd, lsd = O.to_sd(np_unet), O.to_sd(W.make_lpips_state_dict(0))
    coords = O.get_coords(w, h)
    pred = torch.stack([O.predict_lip_image(sd_, coords, win[b], idx[b], h, w, u01[b]) for b in range(B)])
    lip = pred.reshape(B, h, w, 3)
    new = torch.cat([O.composite(lip[b:b + 1], face, gt[b:b + 1], mask, x0, y0, coord[b:b + 1])[0] for b in range(B)])
    recon = O.unet_forward(usd, new)
    ref = (O.mse_loss(pred, targets) + O.perceptual_loss(lsd, lip, targets.reshape(B, h, w, 3), 0.05)
           + O.mse_loss(recon, gt) + O.perceptual_loss(lsd, recon, gt, 0.05))
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 2e-5 * abs(float(ref)), (float(loss), float(ref))
    perc = O.perceptual_loss(lsd, lip, targets.reshape(B, h, w, 3), 0.05) + O.perceptual_loss(lsd, recon, gt, 0.05)
    assert abs(float(aux["loss_perceptual"]) - float(perc)) <= 2e-5 * float(perc) and float(perc) > 1e-4
    for k, v in sd_.items():
        if v.grad is None:
            continue
        scale = float(v.grad.abs().max()) + 1e-12
        err = float((g[k].cpu() - v.grad).abs().max())
        assert err <= 5e-4 * scale + 1e-9, f"{k}: max err {err:.3e} vs scale {scale:.3e}"


def test_trainer_train_stage1_with_perceptual_terms_equals_the_fused_step(lp, dev):
    """Trainer.train_stage1 (training.py:347-574) with use_perceptual_loss on -- LPIPS on the lip (:420-421) and on the fused face
    (:453-456) next to the two MSE terms -- leaves the gradients of the fused StageOneStep(perceptual=...) in .grad."""
    import random
    from tests.test_gpu_parity import make_model
    h, w, FH, FW, x0, y0 = 32, 40, 64, 72, 14, 20
    m = make_model(dev, h, w).train()
    m.load_state_dict({k: T(v) for k, v in W.make_unet_state_dict(0).items()})
    m.post_fusion_unet.eval()
    for p in m.post_fusion_unet.parameters():
        p.requires_grad = False
    rng = np.random.default_rng(11)
    data = {"audio": T(W.synthetic_audio(1, seed=5).astype(np.float32)), "rgb": T(rng.random((1, h, w, 3), dtype=np.float32)),
            "index": torch.tensor([17]), "total_frame": torch.tensor([599]), "coord": T(W.synthetic_warp_coords(1, FH, FW, seed=2)),
            "rgb_face_zero": T(rng.random((1, FH, FW, 3), dtype=np.float32)), "rgb_face_ori": T(rng.random((1, FH, FW, 3), dtype=np.float32)),
            "lip_lefttop_x": x0, "lip_lefttop_y": y0}
    mask = torch.zeros(1, FH, FW, 3)
    mask[:, y0:y0 + h, x0:x0 + w] = 1
    data["mask_lip_canonical"] = mask
    holes = (T(rng.standard_normal((1, FH, FW)).astype(np.float32)), T(rng.standard_normal((1, FH, FW)).astype(np.float32)))
    cfg = {**m.cfg, "training": {**m.cfg["training"], "use_canonical_depth_loss_photo_v2": False, "batch_rays": h * w}}
    tr = s2l.Trainer(m, optimizer=torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.0), cfg=cfg,
                     use_syncloss=False, use_perceptual_loss=True, w_perceptual_loss=0.05, perceptual_loss_fn=lp)
    saved = (torch.rand, torch.randn, random.random)
    fq = [holes[0][:, None].repeat(1, 3, 1, 1), holes[1][:, None].repeat(1, 3, 1, 1)]
    torch.rand = lambda *a, **k: torch.full((1,), 0.42, device=dev)
    torch.randn = lambda *a, **k: fq.pop(0)
    random.random = lambda: 0.9
    try:
        _, loss = tr.train_stage1(data, it=10, seed=0)
    finally:
        torch.rand, torch.randn, random.random = saved
    step = s2l.StageOneStep(m, h, w, precision="fp32", face_loss=True, perceptual=lp, w_perceptual_loss=0.05)
    ref_loss, ref_g, aux = step.loss_and_grads(data["audio"].to(dev), [17], data["rgb"].reshape(1, -1, 3).to(dev), [0.42],
                                               face=dict(rgb_face_canonical=data["rgb_face_zero"].to(dev), rgb_face_gt=data["rgb_face_ori"].to(dev),
                                                         mask_lip_canonical=mask.to(dev), lip_lefttop_x=x0, lip_lefttop_y=y0,
                                                         coord=data["coord"].to(dev), hole_noise=(holes[0].to(dev), holes[1].to(dev))))
    assert abs(float(loss["loss"].detach()) - float(ref_loss)) <= 2e-6 * max(1.0, float(ref_loss))
    assert abs(float(loss["loss_perceptual"]) - float(aux["loss_perceptual"])) <= 1e-6 and float(aux["loss_perceptual"]) > 1e-4
    params = dict(m.named_parameters())
    for k, gk in ref_g.items():
        assert relerr(params[k].grad.reshape(gk.shape), gk.cpu()) <= 2e-4, (k, relerr(params[k].grad.reshape(gk.shape), gk.cpu()))
