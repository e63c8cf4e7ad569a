"""Loops the U-Net's inference forward (16 frames 500 x 500) in one precision for a few seconds (power / clock probing: tools/dev/power_probe.sh).
    python tools/dev/loop_unet_split.py <fp32|split|bf16> [seconds=7]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
prec = sys.argv[1] if len(sys.argv) > 1 else "split"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
dev = torch.device("cuda:0")
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
x = torch.rand(16, 500, 500, 3, device=dev)
u.forward_nhwc(x, precision=prec)
torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        u.forward_nhwc(x, precision=prec)
    e1.record()
    torch.cuda.synchronize()
    last = e0.elapsed_time(e1) / 10
print(f"{prec}: {last:.2f} ms per 16 frames")
