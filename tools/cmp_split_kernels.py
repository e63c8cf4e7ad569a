"""The persistent split-bf16 convolution kernel (s2l_set_unet_split_kernel(0)) against the one-tile-per-workgroup form (1): same bits.
    python tools/cmp_split_kernels.py [frames=3] [H=500] [W=500]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W, _abi
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H = int(sys.argv[2]) if len(sys.argv) > 2 else 500
Wd = int(sys.argv[3]) if len(sys.argv) > 3 else 500
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
x = torch.rand(F, H, Wd, 3, device=dev)
lib = _abi.load()
outs = []
for kind in (1, 0, 0):
    _abi.check(lib.s2l_set_unet_split_kernel(kind), "s2l_set_unet_split_kernel")
    outs.append(u.forward_nhwc(x, precision="split").clone())
torch.cuda.synchronize()
lib.s2l_set_unet_split_kernel(0)
ref = u.forward_nhwc(x)
print("persistent == one-tile form:", torch.equal(outs[0], outs[1]), " run-to-run:", torch.equal(outs[1], outs[2]),
      " max |diff|:", float((outs[0] - outs[1]).abs().max()), " rmse vs fp32:", float(((outs[1] - ref) ** 2).mean().sqrt()))
