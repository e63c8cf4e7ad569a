#!/usr/bin/env python3
"""Per-stage phase times of the bf16 forward kernel (experiment build with -DS2L_TRACE16, loaded through S2L_LIB):
waves 0 and 4 of workgroup 0, first tile: k-loops, epilogue, stage copy to LDS, barrier wait.  Cycles of s_memtime."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
from speech2lip_amd.talking_face import _ptr, _stream
dev = torch.device("cuda:0")
m = s2l.TalkingFace(dev, s2l.may_config(96, 96)).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(0, "he", include_dead=True).items()})
lib = _abi.load()
raw = ctypes.CDLL(os.environ["S2L_LIB"])
N = 4 * 96 * 96 * 16
Np = int(lib.s2l_bf16_rows_padded(N)); lay = Np * 256
x = torch.randn(N, 128, device=dev) * 0.5
xT = torch.empty(Np * 128, dtype=torch.int16, device=dev)
lib.s2l_rows_to_tiles_bf16(_ptr(x), 128, _ptr(xT), N, _stream())
hT = torch.empty(8 * lay, dtype=torch.int16, device=dev)
masks = torch.empty(8 * (Np // 64) * 256, dtype=torch.int64, device=dev)
rgb = torch.empty(N, 3, device=dev)
tr = torch.zeros(2 * 32 * 8, dtype=torch.int64, device=dev)
pb, pf = m.packed_weights_bf16(), m.packed_weights()
for it in range(2):
    raw.s2l_trace16_set(ctypes.c_void_p(tr.data_ptr() if it else 0))
    lib.s2l_train_forward_bf16(_ptr(pb), _ptr(pf), _ptr(xT), _ptr(hT), _ptr(masks), _ptr(rgb), N, _stream())
    torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(2, 32, 8)
for w in range(2):
    d = t[w]
    k = d[:, 1] - d[:, 0]; e = d[:, 2] - d[:, 1]; ls = d[:, 3] - d[:, 2]; bar = d[:, 4] - d[:, 3]
    tot = d[1:, 0] - d[:-1, 0]
    print(f"wave {4 * w}: per stage (cycles): gload+k-loops {k.mean():.0f}  epilogue {e.mean():.0f}  lds store {ls.mean():.0f}  barrier {bar.mean():.0f}  stage {tot.mean():.0f}")
    print("   k-loops by layer:", [int(k[4 * L:4 * L + 4].mean()) for L in range(8)], " epilogue by layer:", [int(e[4 * L:4 * L + 4].mean()) for L in range(8)])
