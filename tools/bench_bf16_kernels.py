#!/usr/bin/env python3
"""Kernel-level timing of the bf16 training kernels (64 frames 96x96 = 2,359,296 rows): forward, backward, one weight
gradient.  S2L_LIB=<variant .so> selects an experiment build.   python tools/bench_bf16_kernels.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from speech2lip_amd.talking_face import _ptr, _stream

_args = [x for x in sys.argv[1:] if not x.startswith("--")]
B = int(_args[0]) if _args else 64
dev = torch.device("cuda:0")
m = s2l.TalkingFace(dev, s2l.may_config(96, 96)).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
lib = _abi.load()
N = 4 * 96 * 96 * B
Np = int(lib.s2l_bf16_rows_padded(N)); lay = Np * 256
torch.manual_seed(0)
x = torch.randn(N, 128, device=dev) * 0.5
xT = torch.empty(Np * 128, dtype=torch.int16, device=dev)
lib.s2l_rows_to_tiles_bf16(_ptr(x), 128, _ptr(xT), N, _stream())
hT = torch.empty(8 * lay, dtype=torch.int16, device=dev); dzT = torch.empty_like(hT)
masks = torch.empty(8 * (Np // 64) * 256, dtype=torch.int64, device=dev)
rgb, drgb, dxa = torch.empty(N, 3, device=dev), torch.randn(N, 3, device=dev) * 1e-3, torch.empty(N, 64, device=dev)
work = torch.empty(int(lib.s2l_wgrad_bf16_work_floats()), device=dev)
dw, db = torch.empty(256, 256, device=dev), torch.empty(256, device=dev)
pb, pf = m.packed_weights_bf16(), m.packed_weights()

if os.environ.get("S2L_FWD_KIND"):      # 1 = the C++ forward kernel instead of the assembly one
    assert lib.s2l_set_bf16_forward_kernel(int(os.environ["S2L_FWD_KIND"])) == 0

def timed(fn, it=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it

tf = timed(lambda: lib.s2l_train_forward_bf16(_ptr(pb), _ptr(pf), _ptr(xT), _ptr(hT), _ptr(masks), _ptr(rgb), N, _stream()))
tb = timed(lambda: lib.s2l_train_backward_bf16(_ptr(pb), _ptr(drgb), _ptr(masks), _ptr(dzT), _ptr(dxa), N, _stream()))
tiles = torch.empty(Np // 256, 64, device=dev)
tba = timed(lambda: lib.s2l_train_backward_bf16_tiles(_ptr(pb), _ptr(drgb), _ptr(masks), _ptr(dzT), _ptr(tiles), N, _stream()))
tw = timed(lambda: lib.s2l_wgrad_bf16(_ptr(dzT[3 * lay:]), _ptr(hT[2 * lay:]), 256, _ptr(work), _ptr(dw), _ptr(db), N, _stream()))
if "--digest" in sys.argv:      # A/B aid: two builds with the same arithmetic print the same digests
    import hashlib
    torch.cuda.synchronize()
    hsh = lambda x: hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()[:12]
    print("digests: hT", hsh(hT), "masks", hsh(masks), "rgb", hsh(rgb), "dzT", hsh(dzT), "dxa", hsh(dxa), "dw", hsh(dw), "db", hsh(db))
gb = 8 * lay * 2 / 1e9
print(f"{os.environ.get('S2L_LIB', 'default'):28s} rows {N}: forward {tf:.3f} ms ({gb / tf * 1e3:.0f} GB/s of tiles written), "
      f"backward {tb:.3f} ms (C++) / {tba:.3f} ms (assembly), one wgrad {tw:.3f} ms ({2 * lay * 2 / 1e9 / tw * 1e3:.0f} GB/s read)")
