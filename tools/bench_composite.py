"""Composite (paste + warp) throughput on MI355X: BASELINE config 3 geometry (128x128 lip in a
500x500 face).  HBM roofline: algorithmic bytes per frame = 8,000,000 + 12*h*w.
    python tools/bench_composite.py [frames]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
h = w = 128; FH = FW = 500; x0, y0 = 186, 300
m = s2l.TalkingFace(dev, s2l.may_config(h, w)).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
g = torch.Generator(device=dev).manual_seed(2)
lip = torch.rand(F, h, w, 3, device=dev, generator=g)
face = torch.rand(1, FH, FW, 3, device=dev, generator=g)
gt = torch.rand(F, FH, FW, 3, device=dev, generator=g)
mask = torch.zeros(1, FH, FW, 3, device=dev); mask[:, y0:y0 + h, x0:x0 + w] = 1
ys, xs = torch.meshgrid(torch.arange(FH, device=dev), torch.arange(FW, device=dev), indexing="ij")
ident = torch.stack([(2 * xs + 1) / FW - 1, (2 * ys + 1) / FH - 1], -1).float()
ang = (torch.rand(F, device=dev, generator=g) - 0.5) * (6 * np.pi / 180)          # rotation <= 3 deg
rot = torch.stack([torch.stack([ang.cos(), -ang.sin()], -1), torch.stack([ang.sin(), ang.cos()], -1)], -2)
shift = (torch.rand(F, 1, 1, 2, device=dev, generator=g) - 0.5) * 0.04
coord = (torch.einsum("hwk,fjk->fhwj", ident, rot) + shift + torch.randn(F, FH, FW, 2, device=dev, generator=g) * 1e-3).clamp(-1, 1).contiguous()
out = torch.empty(F, FH, FW, 3, device=dev)
for _ in range(3):
    m.composite_clip(lip, face, gt, mask, x0, y0, coord, out=out)
torch.cuda.synchronize()
# batches of 10 calls between one event pair, so that the queue stays full and host launch latency is not in the number
evs = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        m.composite_clip(lip, face, gt, mask, x0, y0, coord, out=out)
    e1.record(); evs.append((e0, e1))
torch.cuda.synchronize()
ms = float(np.median([a.elapsed_time(b) for a, b in evs])) / 10
bytes_per_frame = 8_000_000 + 12 * h * w
gbs = bytes_per_frame * F / (ms * 1e-3) / 1e9
# parity spot check on one frame against the CPU oracle
from oracle import s2l_oracle as O
rn, _ = O.composite(lip[:1].cpu(), face.cpu(), gt[:1].cpu(), mask.cpu(), x0, y0, coord[:1].cpu())
err = float((out[:1].cpu() - rn).abs().max())
nbad = int(((out[:1].cpu() - rn).abs() > 1e-5).sum())
print(json.dumps({"kernel": "s2l::composite_kernel", "frames": F, "ms": round(ms, 4), "frames_per_s": round(F / ms * 1e3, 1),
                  "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                               "algorithmic_bytes_per_frame": bytes_per_frame},
                  "parity_frame0": {"max_abs_err": err, "pixels_off_by_more_than_1e-5": nbad}}))
