"""Train-mode half-width U-Net forward + input gradient with the convolutions' own tile statistics (default) against the separate
statistics pass (S2L_NO_CONV_STATS=1): run once per mode (the switch is read once per process), then compare.
    python tools/dev/cmp_conv_stats.py save /tmp/a.pt ; S2L_NO_CONV_STATS=1 python tools/dev/cmp_conv_stats.py save /tmp/b.pt ; python tools/dev/cmp_conv_stats.py cmp /tmp/a.pt /tmp/b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W

if sys.argv[1] == "save":
    dev = torch.device("cuda:0")
    res = {}
    for (F, H, Wd) in ((3, 125, 93), (2, 500, 500), (2, 40, 52), (1, 33, 17)):
        u = s2l.SimpleUnetLight().to(dev).train()
        u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
        for p in u.parameters():
            p.requires_grad_(False)
        g = torch.Generator().manual_seed(H)
        x = torch.rand(F, H, Wd, 3, generator=g).to(dev)
        d = torch.randn(F, H, Wd, 3, generator=g).to(dev)
        out, ctx = u.forward_for_backward(x, precision="bf16")
        dx = u.backward_to_input(ctx, d)
        res[f"{F}x{H}x{Wd}"] = {"out": out.cpu(), "dx": dx.cpu(), "rm": u.inc.double_conv[4].running_mean.cpu().clone(),
                                "rv": u.up1.conv.double_conv[1].running_var.cpu().clone()}
    torch.save(res, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        for f in a[k]:
            x, y = a[k][f].double(), b[k][f].double()
            print(f"{k:12s} {f:4s} rel L2 {float((x - y).norm() / (y.norm() + 1e-30)):.3e}  max abs {float((x - y).abs().max()):.3e}  equal {bool(torch.equal(a[k][f], b[k][f]))}")
