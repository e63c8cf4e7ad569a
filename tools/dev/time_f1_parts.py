"""One frame per call: stream time of the three launches of TalkingFace.render_clip (audio encoder, frame vectors, render kernel), each alone,
back to back (HIP events around 200 calls).  python tools/dev/time_f1_parts.py [size=96]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from speech2lip_amd import _abi, weights as W
from tools.benchlib import make_model
h = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda:0")
lib = _abi.load()
m = make_model(dev, h, h)
a = torch.from_numpy(W.synthetic_audio(1, 1).astype(np.float32)).to(dev)
idx = torch.zeros(1, dtype=torch.int64, device=dev)
out = torch.empty(1, h, h, 3, device=dev)
m.render_clip(a, idx, h, h, out=out)
packed = m.packed_weights()
feat = torch.empty(1, 64, device=dev); q0 = torch.empty(1, 256, device=dev); q5 = torch.empty(1, 256, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timed(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("audio_encode  %.1f us" % timed(lambda: lib.s2l_audio_encode(p(packed), p(a), p(feat), 1, st)))
print("frame_vectors %.1f us" % timed(lambda: lib.s2l_frame_vectors(p(packed), p(feat), p(idx), p(q0), p(q5), 1, st)))
print("render_clip   %.1f us (all three + host)" % timed(lambda: m.render_clip(a, idx, h, h, out=out)))
