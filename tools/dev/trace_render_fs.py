"""Per-phase timestamps of the feature-split tile (render_fs_kernel).  Needs an experiment build with stamps in the assembly body:
    mkdir -p /tmp/fstr && S2L_FS_TRACE=1 python speech2lip_amd/csrc/gen_render_fs_body.py /tmp/fstr
    tools/build_variant.sh render.hip ab/fstrace.so -I/tmp/fstr -DS2L_EXP_TRACE
    python tools/dev/trace_render_fs.py ab/fstrace.so [size=96] [frames=1]
Wave 0 of every workgroup stamps s_memtime at: body start (0), layer L's MFMAs issued (1 + 2 L), layer L's exchange done -- barrier + B reads --
(2 + 2 L), output block done (15), tile end (16); s_memrealtime (100 MHz) at 18 / 17.  The stamps wait for everything in flight through LDS, so the
phases are slightly serialised: read the numbers as an upper bound per phase."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["S2L_LIB"] = os.path.abspath(sys.argv[1])
from speech2lip_amd import _abi, weights as W      # noqa: E402
from tools.benchlib import make_model                # noqa: E402
h = int(sys.argv[2]) if len(sys.argv) > 2 else 96
F = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
lib = _abi.load()
m = make_model(dev, h, h)
a = torch.from_numpy(W.synthetic_audio(F, 1).astype(np.float32)).to(dev)
idx = torch.arange(100, 100 + F, device=dev)
_abi.check(lib.s2l_set_render_shape(4), "shape")
out = torch.empty(F, h, h, 3, device=dev)
for _ in range(5):
    m.render_clip(a, idx, h, h, out=out)
torch.cuda.synchronize()
ntiles = (h * h + 15) // 16 * F
trace = torch.zeros(ntiles * 32, dtype=torch.int64, device=dev)
raw = ctypes.CDLL(os.environ["S2L_LIB"])
raw.s2l_debug_set_trace.argtypes = [ctypes.c_void_p]
assert raw.s2l_debug_set_trace(trace.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); m.render_clip(a, idx, h, h, out=out); e1.record(); torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(ntiles, 32)
n_wg = min(256, ntiles)
first = np.array([ntiles * b // n_wg for b in range(n_wg)])
order = np.zeros(ntiles, dtype=int)
for b in range(n_wg):
    lo, hi = ntiles * b // n_wg, ntiles * (b + 1) // n_wg
    order[lo:hi] = np.arange(hi - lo)
ticks = (t[:, 16] - t[:, 0]).astype(float)
real = (t[:, 17] - t[:, 18]).astype(float)      # 100 MHz
hz = np.median(ticks / np.maximum(real, 1)) * 100e6
print(f"{h}x{h} F={F}: {ntiles} tiles on {n_wg} workgroups, call {e0.elapsed_time(e1) * 1e3:.1f} us; s_memtime runs at {hz / 1e6:.0f} MHz against s_memrealtime")
names = ["h0 (head)"] if False else []
d = np.diff(t[:, :17], axis=1).astype(float)
us = 1e6 / hz
for k in sorted(set(order)):
    sel = order == k
    mf = d[sel][:, 0:14:2]                      # stamps 0 -> 1, 2 -> 3, ...: the layers' MFMA phases
    ex = d[sel][:, 1:14:2]                      # 1 -> 2, ...: ReLU of the last M-block, barrier, B reads
    outb, tail = d[sel][:, 14], d[sel][:, 15]
    tot = ticks[sel]
    print(f"tile #{k} of its workgroup ({int(sel.sum())} tiles): total {np.median(tot) * us:6.2f} us (p10 {np.percentile(tot, 10) * us:.2f}, p90 {np.percentile(tot, 90) * us:.2f})")
    print("   layer MFMA phases  us: " + " ".join(f"{np.median(mf[:, L]) * us:5.2f}" for L in range(7)) + f"   sum {np.median(mf.sum(1)) * us:.2f}")
    print("   layer exchanges    us: " + " ".join(f"{np.median(ex[:, L]) * us:5.2f}" for L in range(7)) + f"   sum {np.median(ex.sum(1)) * us:.2f}")
    print(f"   output block {np.median(outb) * us:.2f} us, store + wait for the next tile's rows {np.median(tail) * us:.2f} us")
start = t[first, 0].astype(float)
end = t[:, 16].astype(float)
print(f"first stamp spread over workgroups {(start.max() - start.min()) * us:.2f} us; first stamp -> last stamp of the launch {(end.max() - start.min()) * us:.2f} us")
print(f"ideal: 256 MFMAs x 32 cycles per layer = {8192 / 2.4e3:.2f} us at 2.4 GHz")
