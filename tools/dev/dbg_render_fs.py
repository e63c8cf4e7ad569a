"""Feature-split tile (s2l_set_render_shape(4)) against the single shape: bits and time.  python tools/dev/dbg_render_fs.py [size=64] [frames=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from speech2lip_amd import _abi, weights as W
from tools.benchlib import make_model
h = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
lib = _abi.load()
m = make_model(dev, h, h)
a = torch.from_numpy(W.synthetic_audio(F, 1).astype(np.float32)).to(dev)
idx = torch.arange(100, 100 + F, device=dev)
def run(mode):
    _abi.check(lib.s2l_set_render_shape(mode), "shape")
    out = torch.full((F, h, h, 3), float("nan"), device=dev)
    m.render_clip(a, idx, h, h, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        m.render_clip(a, idx, h, h, out=out)
    e1.record(); torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / 50 * 1e3
ref, t3 = run(3)
got, t4 = run(4)
lib.s2l_set_render_shape(0)
d = (got - ref)
bad = ~(got == ref)
print(f"{h}x{h} F={F}: single {t3:.1f} us, feature-split {t4:.1f} us per call; equal={bool(torch.equal(got, ref))} nan={int(torch.isnan(got).sum())} "
      f"mismatches={int(bad.sum())} max|d|={float(d.abs().nan_to_num(9e9).max()):.3e}")
if bad.any():
    nz = bad.any(-1).nonzero()[:8].tolist()
    print("first mismatching (frame,y,x):", nz)
    print(got[bad][:6].tolist(), ref[bad][:6].tolist())
