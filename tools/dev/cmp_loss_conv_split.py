#!/usr/bin/env python3
"""Split-operand form of the loss nets' convolutions against the exact fp32 form: SyncNet embeddings / loss / window gradient and the
LPIPS distance / image gradient on seeded inputs.  python tools/dev/cmp_loss_conv_split.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import speech2lip_amd as s2l
from speech2lip_amd import weights as W


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30)), float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    dev = torch.device("cuda:0")
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
    for B in (1, 3, 16):
        mel, pos, neg = (torch.from_numpy(x).to(dev) for x in W.synthetic_sync_batch(B, seed=1))
        out = {}
        for prec in ("fp32", "split"):
            out[prec] = s2l.SyncLoss(net, precision=prec).get_sync_contrastive_loss(mel, pos, neg, want_grad=True)
        (l0, g0), (l1, g1) = out["fp32"], out["split"]
        print(f"syncnet B={B}: loss {float(l0):.7f} vs {float(l1):.7f}; grad rel-l2 / rel-max {rel(g1, g0)}")
    lp = s2l.LPIPS(pretrained=False).to(dev)
    torch.manual_seed(0)
    for N, H, Wd in ((2, 96, 96), (3, 64, 80), (1, 500, 500)):
        a, b = torch.rand(N, H, Wd, 3, device=dev), torch.rand(N, H, Wd, 3, device=dev)
        res = {}
        for prec in ("fp32", "split"):
            d, st = lp.distance_nhwc(a, b, from01=True, keep=True, precision=prec)
            g = lp.backward_nhwc(st, torch.ones(N, device=dev))
            res[prec] = (d.clone(), g.clone())
        print(f"lpips {N}x{H}x{Wd}: d {res['fp32'][0].tolist()} vs {res['split'][0].tolist()}; grad rel {rel(res['split'][1], res['fp32'][1])}")


if __name__ == "__main__":
    main()


def sensitivity():
    """How much the EXACT form's gradient moves when its input moves by 1e-6 relative (ReLU decisions near zero flip): the scale the
    split form's differences have to be read against."""
    dev = torch.device("cuda:0")
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
    sl = s2l.SyncLoss(net)
    for B in (1, 16):
        mel, pos, neg = (torch.from_numpy(x).to(dev) for x in W.synthetic_sync_batch(B, seed=1))
        l0, g0 = sl.get_sync_contrastive_loss(mel, pos, neg, want_grad=True)
        g0 = g0.clone()
        torch.manual_seed(1)
        for eps in (1e-7, 1e-6, 1e-5):
            l1, g1 = sl.get_sync_contrastive_loss(mel, pos * (1 + eps * torch.randn_like(pos)), neg, want_grad=True)
            print(f"syncnet B={B}: exact form, input perturbed by {eps:g}: grad rel {rel(g1, g0)}")


if len(sys.argv) > 1 and sys.argv[1] == "sens":
    sensitivity()
