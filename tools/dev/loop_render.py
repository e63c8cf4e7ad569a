"""Loops the lip renderer (1000 frames 96 x 96) for a few seconds (power / clock probing).  python tools/dev/loop_render.py <fp32|split> [seconds=7] [shape mode 0..4 = s2l_set_render_shape]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from speech2lip_amd import weights as W
from tools.benchlib import make_model
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
dev = torch.device("cuda:0")
if len(sys.argv) > 3:
    from speech2lip_amd import _abi
    _abi.check(_abi.load().s2l_set_render_shape(int(sys.argv[3])), "s2l_set_render_shape")
m = make_model(dev, 96, 96)
a = torch.from_numpy(W.synthetic_audio(1000, 1).astype(np.float32)).to(dev)
i = torch.arange(1000, device=dev)
out = torch.empty(1000, 96, 96, 3, device=dev)
m.render_clip(a, i, 96, 96, out=out, precision=prec)
torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        m.render_clip(a, i, 96, 96, out=out, precision=prec)
    e1.record()
    torch.cuda.synchronize()
    last = e0.elapsed_time(e1) / 5
print(f"{prec}: {last:.2f} ms per 1000 frames")
