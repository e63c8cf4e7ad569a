"""conv16_asm_kernel (generated assembly) vs conv3x3_split_kernel (C++): the U-Net's split-mode forward, bit for bit, and speed.
    python tools/cmp_conv16.py [frames=16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W, _abi
dev = torch.device("cuda:0")
_abi.reference_kernels().__enter__()      # the non-default forms live in libs2l_hip_ref.so
lib = _abi.load()
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
shapes = [(1, 16, 16), (2, 20, 36), (1, 33, 47), (3, 64, 64), (1, 131, 77), (2, 500, 500)]
if len(sys.argv) > 2:
    shapes = shapes[:int(sys.argv[2])]
bad = 0
for (F, H, Wd) in shapes:
    x = torch.rand(F, H, Wd, 3, device=dev)
    lib.s2l_set_unet_split_kernel(0)
    a = u.forward_nhwc(x, precision="split").clone()
    lib.s2l_set_unet_split_kernel(2)
    b = u.forward_nhwc(x, precision="split").clone()
    b2 = u.forward_nhwc(x, precision="split").clone()
    torch.cuda.synchronize()
    eq, det = bool(torch.equal(a, b)), bool(torch.equal(b, b2))
    bad += not (eq and det)
    print(F, H, Wd, "asm == C++:", eq, "deterministic:", det, "max diff", float((a - b).abs().max()), flush=True)
lib.s2l_set_unet_split_kernel(0)
Fb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.rand(Fb, 500, 500, 3, device=dev)
for kind in (0, 2):
    lib.s2l_set_unet_split_kernel(kind)
    y = u.forward_nhwc(x, precision="split")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        u.forward_nhwc(x, out=y, precision="split")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("kind", kind, f"{dt*1e3:.2f} ms per {Fb} frames = {Fb/dt:.0f} frames/s", flush=True)
lib.s2l_set_unet_split_kernel(0)
sys.exit(1 if bad else 0)
