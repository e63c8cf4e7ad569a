"""Loops ONE layer of the half-width convolution for a few seconds (power / clock probing: tools/dev/power_probe.sh).
    python tools/dev/loop_convh_layer.py <layer 1..9> <transposed 0|1> [seconds=6]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import _abi, weights as W
CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128), (256, 128), (128, 64), (128, 64), (64, 64)]
LVL = [0, 0, 1, 1, 2, 2, 1, 1, 0, 0]
l, tr = int(sys.argv[1]), int(sys.argv[2])
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
F, S = 20, 500
p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
dev = torch.device("cuda:0")
lib = _abi.load()
u = s2l.SimpleUnetLight().to(dev).train()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
tensors = u._tensors()
raw, raw16 = u._raw_blobs(tensors, u._table(tensors), True)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
cin, cout = CONVS[l]
if tr:
    cin, cout = cout, cin
h = S >> LVL[l]
cat = l in (6, 8) and not tr
CA, CB = (cin // 2, cin // 2) if cat else (cin, 0)
a = torch.randn(F + 1, h, h, CA, device=dev).to(torch.bfloat16)[:F]
b = torch.randn(F + 1, h, h, CB, device=dev).to(torch.bfloat16)[:F] if CB else None
out = torch.empty(F, h, h, cout, dtype=torch.int16, device=dev)
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        _abi.check(lib.s2l_convh_layer(p(raw16), l, tr, p(a), CA, p(b), CB, None, p(out), h, h, F, st), "convh")
    e1.record()
    torch.cuda.synchronize()
    n += 1
    last = e0.elapsed_time(e1) / 50 * 1e3
print(f"layer {l} tr {tr}: {last:.0f} us per launch")
