import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W, _abi
dev = torch.device("cuda:0"); lib = _abi.load()
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
x = torch.rand(16, 500, 500, 3, device=dev)
lib.s2l_set_unet_split_kernel(2)
y = u.forward_nhwc(x, precision="split"); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): u.forward_nhwc(x, out=y, precision="split")
torch.cuda.synchronize()
print(os.environ.get("S2L_LIB", "default"), f"{(time.perf_counter()-t0)/5*1e3:.2f} ms per 16 frames")
