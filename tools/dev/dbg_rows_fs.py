"""rgb_forward's feature-split tile (s2l_set_rows_kernel(2)) against the column form: bits and time.  python tools/dev/dbg_rows_fs.py [rows=4096]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech2lip_amd import _abi
from tools.benchlib import make_model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
lib = _abi.load()
m = make_model(dev, 64, 64)
g = torch.Generator(device="cpu").manual_seed(n)
rows = torch.cat([torch.rand(n, 2, generator=g) * 2 - 1, torch.randn(n, 64, generator=g)], -1).to(dev)
t = torch.tensor([4321], device=dev)
def run(kind):
    _abi.check(lib.s2l_set_rows_kernel(kind), "kind")
    with torch.no_grad():
        out = m.rgb_forward(rows, time_pts=t).clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            m.rgb_forward(rows, time_pts=t)
        e1.record(); torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / 50 * 1e3
ref, t1 = run(1)
got, t2 = run(2)
lib.s2l_set_rows_kernel(0)
bad = ~(got == ref)
print(f"{n} rows: column form {t1:.1f} us, feature-split {t2:.1f} us per rgb_forward call; equal={bool(torch.equal(got, ref))} nan={int(torch.isnan(got).sum())} "
      f"mismatching rows={int(bad.any(-1).sum())} max|d|={float((got - ref).abs().nan_to_num(9e9).max()):.3e}")
if bad.any():
    print("first mismatching rows:", bad.any(-1).nonzero()[:12].flatten().tolist())
    print(got[bad.any(-1)][:4].tolist(), ref[bad.any(-1)][:4].tolist())
