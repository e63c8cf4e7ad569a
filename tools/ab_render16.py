"""Split-bf16 render mode vs the exact fp32 kernel and the CPU oracle: accuracy at small sizes, speed at 96x96x1000.
    python tools/ab_render16.py [frames=1000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
from oracle import s2l_oracle as O
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
def model(h, w):
    m = s2l.TalkingFace(dev, s2l.may_config(h, w), mode="eval").eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
    return m
sd = O.to_sd(W.make_state_dict(0, "he"))
for (h, w, f) in [(16, 16, 12), (12, 20, 5), (5, 7, 1), (24, 24, 30)]:
    m = model(h, w)
    win = torch.from_numpy(W.synthetic_audio(f, seed=1).astype(np.float32))
    idx = list(range(3, 3 + f))
    a = m.render_clip(win.to(dev), idx, h, w)
    b = m.render_clip(win.to(dev), idx, h, w, precision="split")
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.render_clip(sd, win, idx, h, w)
    print(f"{h}x{w}x{f}: fp32 vs oracle rmse {O.rmse(a.cpu(), ref):.2e} | split vs oracle rmse {O.rmse(b.cpu(), ref):.2e} max {float((b.cpu()-ref).abs().max()):.2e} "
          f"psnr {O.psnr(b.cpu(), ref):.1f} | split vs fp32 rmse {O.rmse(b.cpu(), a.cpu()):.2e}", flush=True)
h = w = 96
m = model(h, w)
win = torch.from_numpy(W.synthetic_audio(F, seed=1).astype(np.float32)).to(dev)
idx = torch.arange(F, device=dev)
for prec in ("fp32", "split"):
    out = m.render_clip(win, idx, h, w, precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        m.render_clip(win, idx, h, w, out=out, precision=prec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{prec}: {dt*1e3:.2f} ms per {F} frames = {F/dt:.0f} frames/s", flush=True)
a = m.render_clip(win, idx, h, w)
b = m.render_clip(win, idx, h, w, precision="split")
b2 = m.render_clip(win, idx, h, w, precision="split")
print("96x96 split vs fp32 rmse", O.rmse(b.cpu(), a.cpu()), "max", float((a - b).abs().max()), "deterministic", bool(torch.equal(b, b2)))
with torch.no_grad():
    ref = O.render_clip(sd, win[:2].cpu(), [0, 1], h, w)
print("96x96 frames 0,1 split vs oracle rmse", O.rmse(b[:2].cpu(), ref), "psnr", O.psnr(b[:2].cpu(), ref))
