"""StageOneStep.loss_and_grads twice on the same inputs and the same weights: which outputs are not the same bits?
    python tools/dev/dbg_stage_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
from tools import benchlib
from tools.benchlib import make_model, sync_batch, device_warp_coords
dev = torch.device("cuda:0")
H = Wd = 96
B = 2
for label, kw in (("lip only", dict()), ("face (frozen, train BN)", dict(face=True)), ("face + sync", dict(face=True, sync=True)),
                  ("face, U-Net trains", dict(face=True, early=True))):
    m = make_model(dev, H, Wd, unet=True, train=True)
    if not kw.get("early"):
        for p in m.post_fusion_unet.parameters():
            p.requires_grad = False
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
    step = s2l.StageOneStep(m, H, Wd, syncnet=net if kw.get("sync") else None, precision="bf16", face_loss=bool(kw.get("face")))
    audio = torch.from_numpy(W.synthetic_audio(B, 1).astype(np.float32)).to(dev)
    target = torch.rand(B, H * Wd, 3, device=dev)
    sync = sync_batch(dev, B)
    coord, g = device_warp_coords(dev, B, seed=5)
    face = dict(rgb_face_canonical=sync["rgb_face_canonical"], rgb_face_gt=sync["rgb_face_gt"], mask_lip_canonical=sync["mask_lip_canonical"],
                lip_lefttop_x=sync["lip_lefttop_x"], lip_lefttop_y=sync["lip_lefttop_y"], coord=coord,
                hole_noise=(torch.randn(B, 500, 500, device=dev, generator=g), torch.randn(B, 500, 500, device=dev, generator=g)))
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    ref = None
    for rep in range(4):
        m.load_state_dict(sd0)      # (the train-mode BatchNorm moves its running statistics)
        loss, gr, aux = step.loss_and_grads(audio, list(range(B)), target, [0.5] * B, sync=sync if kw.get("sync") else None,
                                            face=face if kw.get("face") else None)
        torch.cuda.synchronize()
        cur = {k: v.clone() for k, v in gr.items()}
        cur["__loss"] = loss.clone()
        if ref is None:
            ref = cur
        else:
            bad = [(k, float((cur[k] - ref[k]).abs().max() / ref[k].abs().max())) for k in ref if not torch.equal(cur[k], ref[k])]
            print(label, "rep", rep, "differs:", bad[:5], len(bad))
