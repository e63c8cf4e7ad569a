"""Which host-side calls issue the device-to-device copies of one early-phase StageOneStep iteration?  (torch.profiler with stacks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
from tools.benchlib import make_model, sync_batch, device_warp_coords
dev = torch.device("cuda:0")
H = Wd = 96
B = 8
m = make_model(dev, H, Wd, unet=True, train=True)
lp = s2l.LPIPS(pretrained=False).to(dev)
lp.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_lpips_state_dict(0).items()})
opt = torch.optim.Adam([p for n_, p in m.named_parameters() if not n_.startswith("coord_linears")], lr=1e-4)
step = s2l.StageOneStep(m, H, Wd, syncnet=None, precision="bf16", face_loss=True, perceptual=lp)
audio = torch.from_numpy(W.synthetic_audio(B, 1).astype(np.float32)).to(dev)
target = torch.rand(B, H * Wd, 3, device=dev)
sync = sync_batch(dev, B)
coord, g = device_warp_coords(dev, B, seed=5)
face = dict(rgb_face_canonical=sync["rgb_face_canonical"], rgb_face_gt=sync["rgb_face_gt"], mask_lip_canonical=sync["mask_lip_canonical"],
            lip_lefttop_x=sync["lip_lefttop_x"], lip_lefttop_y=sync["lip_lefttop_y"], coord=coord,
            hole_noise=(torch.randn(B, 500, 500, device=dev, generator=g), torch.randn(B, 500, 500, device=dev, generator=g)))
def one():
    loss, gr, aux = step.loss_and_grads(audio, list(range(B)), target, [0.5] * B, face=face)
    s2l.training.apply_grads(m, gr)
    opt.step()
for _ in range(3):
    one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    one()
    torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for ev in prof.events():
    n = ev.name
    if n.startswith("aten::") and n in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::to", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::add_", "aten::zeros_like", "aten::full"):
        st = [s for s in (ev.stack or []) if "speech2lip_amd" in s or "benchlib" in s]
        cnt[(n, st[0] if st else "?")] += 1
for (n, s), c in cnt.most_common(40):
    print(c, n, s[-110:])
