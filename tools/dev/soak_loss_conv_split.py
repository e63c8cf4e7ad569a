"""Random image sizes and batches: the split-operand form of the loss nets' convolutions against the exact form (LPIPS distance, the
gradient kernels over the exact forward's activations; SyncNet embeddings and window gradient over the exact forward's activations).
    python tools/dev/soak_loss_conv_split.py [rounds=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
from speech2lip_amd.syncnet import sync_window
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(7)
torch.manual_seed(11)
lp = s2l.LPIPS(pretrained=False).to(dev)
lp.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_lpips_state_dict(0).items()})
net = s2l.SyncNet_color().to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
worst = {"lpips_d": 0.0, "lpips_g": 0.0, "sync_e": 0.0, "sync_g": 0.0}
for r in range(rounds):
    N, H, Wd = int(rng.integers(1, 5)), int(rng.integers(31, 260)), int(rng.integers(31, 260))
    a, b = torch.rand(N, H, Wd, 3, device=dev), torch.rand(N, H, Wd, 3, device=dev)
    d0, st0 = lp.distance_nhwc(a, b, from01=True, keep=True)
    d1, _ = lp.distance_nhwc(a, b, from01=True, keep=True, precision="split")
    w = torch.rand(N, device=dev)
    g0 = lp.backward_nhwc(st0, w)
    g1 = lp.backward_nhwc(st0[:-1] + (True,), w)
    worst["lpips_d"] = max(worst["lpips_d"], float(((d1 - d0).abs() / d0.abs().clamp_min(1e-6)).max()))
    worst["lpips_g"] = max(worst["lpips_g"], rel(g1, g0))
    assert torch.isfinite(d1).all() and torch.isfinite(g1).all(), (N, H, Wd)
    B = int(rng.integers(1, 20))
    mel, pos, neg = (torch.from_numpy(x).to(dev) for x in W.synthetic_sync_batch(B, seed=100 + r))
    face = torch.cat([sync_window(pos), sync_window(neg)])
    a1, v1 = (t.clone() for t in net.embed_pair_nhwc(mel, face, precision="split"))
    a0, v0 = (t.clone() for t in net.embed_pair_nhwc(mel, face, precision="fp32"))
    dd = torch.nn.functional.normalize(torch.randn(B, 512, device=dev), dim=1)
    s0 = net.face_backward(dd).clone()
    net._last = net._last[:2] + (True,)
    s1 = net.face_backward(dd)
    worst["sync_e"] = max(worst["sync_e"], float((v1 - v0).abs().max()), float((a1 - a0).abs().max()))
    worst["sync_g"] = max(worst["sync_g"], rel(s1, s0))
print(f"{rounds} rounds: worst relative LPIPS distance {worst['lpips_d']:.2e}, LPIPS gradient kernels {worst['lpips_g']:.2e}, "
      f"SyncNet embeddings (abs) {worst['sync_e']:.2e}, SyncNet gradient kernels {worst['sync_g']:.2e}")
