"""Experiment: per-workgroup phase timestamps of the render kernel (needs ubin/exp_TRACE.so)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2L_LIB"] = os.path.abspath(sys.argv[1])
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
dev = torch.device("cuda:0")
H = Wd = 96; F = 200
m = s2l.TalkingFace(dev, s2l.may_config(H, Wd)).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
audio = torch.from_numpy(W.synthetic_audio(F, 1).astype(np.float32)).to(dev)
m.render_clip(audio, list(range(F)), H, Wd); torch.cuda.synchronize()
ntiles = F * H * Wd // 192
trace = torch.zeros(ntiles * 16, dtype=torch.int64, device=dev)
lib = ctypes.CDLL(os.environ["S2L_LIB"])
lib.s2l_debug_set_trace.argtypes = [ctypes.c_void_p]
assert lib.s2l_debug_set_trace(trace.data_ptr()) == 0
m.render_clip(audio, list(range(F)), H, Wd); torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(ntiles, 16)
names = ["prologue(loads)", "wait slab0+barrier", "layer0", "layer1", "layer2", "layer3", "layer4", "layer5", "layer6", "out layer+store", "dma drain"]
d = np.diff(t[:, :12], axis=1)
print("tiles", ntiles, " per-phase cycles (s_memtime ticks): median / p10 / p90")
for i, n in enumerate(names):
    print(f"  {n:22s} {np.median(d[:, i]):10.0f} {np.percentile(d[:, i], 10):10.0f} {np.percentile(d[:, i], 90):10.0f}")
wall = (t[:, 15] - t[:, 14]).astype(np.float64)   # s_memrealtime ticks (constant 100 MHz)
tot = t[:, 11] - t[:, 0]
print("  s_memtime ticks per s_memrealtime tick: median %.3f  => s_memtime rate %.1f MHz if realtime is 100 MHz" % (np.median(tot / wall), 100 * np.median(tot / wall)))
print("  tile wall time (us @100MHz): median %.2f" % (np.median(wall) / 100.0))
print("  whole kernel span (ms @100MHz): %.3f" % ((t[:, 15].max() - t[:, 14].min()) / 1e5))
print("  total in-kernel per tile", np.median(tot), "span of whole kernel", t[:, 11].max() - t[:, 0].min())
# gaps between consecutive tiles on the same CU (hw_id: cu/se/xcc)
key = t[:, 13] * (1 << 32) + (t[:, 12] & 0xFFFFFF00 & ~0xF0)  # coarse: xcc + hwid sans wave/simd bits
order = np.argsort(t[:, 0])
last_end = {}
gaps = []
for i in order:
    k = (int(t[i, 13]), int(t[i, 12]) >> 8 & 0xF, int(t[i, 12]) >> 13 & 0x7, int(t[i,12]) >> 16 & 0x1)  # xcc, cu_id, se_id, sh
    if k in last_end:
        gaps.append(t[i, 0] - last_end[k])
    last_end[k] = t[i, 11]
gaps = np.array(gaps)
print("distinct CU keys", len(last_end), " inter-tile gap cycles: median", np.median(gaps), "p10", np.percentile(gaps, 10), "p90", np.percentile(gaps, 90))
