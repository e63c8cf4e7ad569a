"""The face term's pieces twice on the same inputs: composite -> U-Net (train-mode BatchNorm, bf16 chain) -> MSE -> backward -> composite
adjoint.  Which tensor is not the same bits?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi
from tools.benchlib import make_model, sync_batch, device_warp_coords
dev = torch.device("cuda:0")
H = Wd = 96
B = 2
m = make_model(dev, H, Wd, unet=True, train=True)
for p in m.post_fusion_unet.parameters():
    p.requires_grad = False
sync = sync_batch(dev, B)
coord, g = device_warp_coords(dev, B, seed=5)
holes = (torch.randn(B, 500, 500, device=dev, generator=g), torch.randn(B, 500, 500, device=dev, generator=g))
lip = torch.rand(B, H, Wd, 3, device=dev)
gt = sync["rgb_face_gt"].to(dev).float()
args = (sync["rgb_face_canonical"], sync["mask_lip_canonical"], sync["lip_lefttop_x"], sync["lip_lefttop_y"], coord)
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
lib = _abi.load()
ref = None
for prec in ("bf16", "fp32"):
  ref = None
  for rep in range(5):
    m.load_state_dict(sd0)
    new, _ = m.composite_clip(lip, args[0], gt, args[1], args[2], args[3], args[4], hole_noise=holes)
    recon, saved = m.post_fusion_unet.forward_for_backward(new, precision=prec)
    d_recon = (recon - gt) * 1e-3
    d_new = m.post_fusion_unet.backward_to_input(saved, d_recon, None)
    d_lip = m.composite_backward_lip(d_new, args[0], args[1], args[2], args[3], args[4], H, Wd, hole_noise=holes)
    torch.cuda.synchronize()
    cur = dict(new=new.clone(), recon=recon.clone(), d_new=d_new.clone(), d_lip=d_lip.clone())
    if ref is None:
        ref = cur
    else:
        print(prec, "rep", rep, {k: (float((cur[k] - ref[k]).abs().max()), int((cur[k] != ref[k]).sum())) for k in ref})
