"""Post-fusion U-Net throughput at the reference's 500x500 face frame.
FLOPs per frame: 2 * 78.7 GMAC (ten 3x3 convs + 1x1).   python tools/bench_unet.py [frames=16]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = Wd = 500
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
x = torch.rand(F, H, Wd, 3, device=dev)
out = torch.empty_like(x)
for _ in range(2):
    u.forward_nhwc(x, out=out)
torch.cuda.synchronize()
evs = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); u.forward_nhwc(x, out=out); e1.record(); evs.append((e0, e1))
torch.cuda.synchronize()
ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
macs = 0
for (name, cin, cout), (h, w) in zip(W.UNET_CONVS, [(500, 500)] * 2 + [(250, 250)] * 2 + [(125, 125)] * 2 + [(250, 250)] * 2 + [(500, 500)] * 2):
    macs += cin * cout * 9 * h * w
macs += 64 * 3 * 500 * 500
tf = 2 * macs * F / (ms * 1e-3) / 1e12
print(json.dumps({"kernel": "s2l_unet_forward (conv3x3_kernel + ...)", "frames": F, "ms": round(ms, 3),
                  "frames_per_s": round(F / ms * 1e3, 1), "gflop_per_frame": round(2 * macs / 1e9, 2),
                  "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4)}}))
