import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
T = torch.from_numpy
dev = torch.device("cuda:0")
def net():
    u = s2l.SimpleUnetLight().to(dev).train()
    u.load_state_dict({k[len("post_fusion_unet."):]: T(v) for k, v in W.make_unet_state_dict(0).items()})
    return u
rel = lambda a, b: float((a - b).norm() / b.norm())
cos = lambda a, b: float((a.flatten() @ b.flatten()) / (a.norm() * b.norm()))
for (F, fh, fw, kind) in [(2, 64, 80, "noise"), (2, 500, 500, "noise"), (3, 40, 56, "ones"), (2, 500, 500, "smooth")]:
    x = T(W.synthetic_image((F, fh, fw, 3), 5, "x")).to(dev)
    if kind == "noise":
        d = T(np.random.default_rng(2).standard_normal((F, fh, fw, 3)).astype(np.float32)).to(dev)
    elif kind == "ones":
        d = torch.ones(F, fh, fw, 3, device=dev)
    else:
        d = (x - 0.5) * 2e-5      # like an MSE gradient against a nearby target
    r = {}
    for prec in ("fp32", "bf16", "bf16h"):
        u = net()
        o, c = u.forward_train_frames_nhwc(x, precision=prec)
        r[prec] = (o, u.backward_train_frames(c, d))
    for prec in ("bf16", "bf16h"):
        print(f"{F}x{fh}x{fw} {kind:6s} {prec:6s}: out rel {rel(r[prec][0], r['fp32'][0]):.3e} cos {cos(r[prec][0], r['fp32'][0]):.6f} | dx rel {rel(r[prec][1], r['fp32'][1]):.3f} cos {cos(r[prec][1], r['fp32'][1]):.4f}", flush=True)
    print(f"   bf16h vs bf16: dx cos {cos(r['bf16h'][1], r['bf16'][1]):.4f}")
